/*
 * Camera motion of the reference host, src/main.rs:45-142 (pure host code, f32 throughout).
 *
 * keys = [W, S, D, A, Space, ShiftLeft, I, K, L, J, O, U] (main.rs:39, 710-721).  Per tick
 * (main.rs:674) the pressed keys add +-CAMERA_SPEED to a camera-local movement vector and
 * +-CAMERA_ROT_SPEED degrees to a camera-local rotation vector (:51-102); both are turned into the
 * global frame with the host's rotate_vector (:117-142), using the rotation BEFORE this tick for
 * both (:105, :111), and added to position and rotation (:106-114).  Any pressed key restarts the
 * accumulation (frame_count = 0, :47-49).
 *
 * Arithmetic: Rust f32, left to right, no contraction; to_radians() is one f32 multiplication by
 * the f32 constant PI/180 (core::f32::to_radians); cos/sin are the platform's f32 libm (cosf/sinf),
 * which is also what this file calls.
 */
#include "../../include/pt_api.h"

#include <cmath>

namespace {

const float kRadsPerDeg = 3.14159274101257324f / 180.0f;      /* consts::PI / 180.0 in f32 */

void rotate_vector(const float v[3], const float rot_deg[3], float out[3]) {
    const float x = v[0], y = v[1], z = v[2];
    const float a = rot_deg[0] * kRadsPerDeg, b = rot_deg[1] * kRadsPerDeg, c = rot_deg[2] * kRadsPerDeg;
    const float cos_a = cosf(a), sin_a = sinf(a), cos_b = cosf(b), sin_b = sinf(b), cos_c = cosf(c), sin_c = sinf(c);
    /* main.rs:133-139, Rust precedence: binary * before +, -x is (-x) */
    out[0] = x * cos_c * cos_b + y * (cos_c * sin_b * sin_a - sin_c * cos_a) + z * (cos_c * sin_b * cos_a + sin_c * sin_a);
    out[1] = x * sin_c * cos_b + y * (sin_c * sin_b * sin_a + cos_c * cos_a) + z * (sin_c * sin_b * cos_a - cos_c * sin_a);
    out[2] = -x * sin_b + y * cos_b * sin_a + z * cos_b * cos_a;
}

}  // namespace

extern "C" int pt_host_rotate_vector(const float v[3], const float rotation_degrees[3], float out[3]) {
    if (!v || !rotation_degrees || !out) return PT_EINVAL;
    float r[3];
    rotate_vector(v, rotation_degrees, r);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
    return PT_OK;
}

extern "C" int pt_update_camera(const uint8_t keys[12], float speed, float rot_speed,
                                float position[3], float rotation_degrees[3], int *any_key) {
    if (!keys || !position || !rotation_degrees) return PT_EINVAL;
    bool any = false;
    for (int i = 0; i < 12; i++) any = any || keys[i];
    if (any_key) *any_key = any ? 1 : 0;
    float mv[3] = {0.0f, 0.0f, 0.0f}, rt[3] = {0.0f, 0.0f, 0.0f};
    if (keys[0]) mv[2] -= speed;
    if (keys[1]) mv[2] += speed;
    if (keys[2]) mv[0] += speed;
    if (keys[3]) mv[0] -= speed;
    if (keys[4]) mv[1] += speed;
    if (keys[5]) mv[1] -= speed;
    if (keys[6]) rt[0] += rot_speed;
    if (keys[7]) rt[0] -= rot_speed;
    if (keys[8]) rt[1] += rot_speed;
    if (keys[9]) rt[1] -= rot_speed;
    if (keys[10]) rt[2] += rot_speed;
    if (keys[11]) rt[2] -= rot_speed;
    float gm[3], gr[3];
    rotate_vector(mv, rotation_degrees, gm);
    rotate_vector(rt, rotation_degrees, gr);       /* still the rotation before this tick */
    for (int i = 0; i < 3; i++) position[i] += gm[i];
    for (int i = 0; i < 3; i++) rotation_degrees[i] += gr[i];
    return PT_OK;
}
