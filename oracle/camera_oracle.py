"""CPU restatement of the reference host's camera motion.  TEST INFRASTRUCTURE ONLY.

Follows src/main.rs:45-142 statement by statement in numpy float32 (every product and sum is
rounded to f32, left to right, as Rust evaluates f32 expressions).  PARITY UNPINNED for the
trigonometric values: the reference calls the platform's f32 cos/sin, whose last bit is not
specified; here they are numpy's float32 cos/sin.  Everything else (key table, signs, which
rotation is used for which step, the accumulation reset) is pinned by the source text alone.
"""
import numpy as np

F = np.float32
KEYS = ("W", "S", "D", "A", "Space", "Shift", "I", "K", "L", "J", "O", "U")    # main.rs:39
RADS_PER_DEG = F(np.float32(np.pi) / F(180.0))                                   # f32::to_radians


def rotate_vector(v, rot_deg, cos=np.cos, sin=np.sin):
    """main.rs:117-142"""
    x, y, z = (F(t) for t in v)
    a, b, c = (F(F(t) * RADS_PER_DEG) for t in rot_deg)
    cos_a, sin_a, cos_b, sin_b, cos_c, sin_c = F(cos(a)), F(sin(a)), F(cos(b)), F(sin(b)), F(cos(c)), F(sin(c))
    x_rot = x * cos_c * cos_b + y * (cos_c * sin_b * sin_a - sin_c * cos_a) + z * (cos_c * sin_b * cos_a + sin_c * sin_a)
    y_rot = x * sin_c * cos_b + y * (sin_c * sin_b * sin_a + cos_c * cos_a) + z * (sin_c * sin_b * cos_a - cos_c * sin_a)
    z_rot = -x * sin_b + y * cos_b * sin_a + z * cos_b * cos_a
    return [F(x_rot), F(y_rot), F(z_rot)]


def update_camera(keys, position, rotation_deg, speed=1.0, rot_speed=1.0, cos=np.cos, sin=np.sin):
    """main.rs:45-115.  keys: set of names from KEYS.  Returns (position, rotation, any_key)."""
    pressed = [k in keys for k in KEYS]
    s, r = F(speed), F(rot_speed)
    mv = [F(0), F(0), F(0)]
    rt = [F(0), F(0), F(0)]
    if pressed[0]: mv[2] = F(mv[2] - s)      # noqa: E701
    if pressed[1]: mv[2] = F(mv[2] + s)      # noqa: E701
    if pressed[2]: mv[0] = F(mv[0] + s)      # noqa: E701
    if pressed[3]: mv[0] = F(mv[0] - s)      # noqa: E701
    if pressed[4]: mv[1] = F(mv[1] + s)      # noqa: E701
    if pressed[5]: mv[1] = F(mv[1] - s)      # noqa: E701
    if pressed[6]: rt[0] = F(rt[0] + r)      # noqa: E701
    if pressed[7]: rt[0] = F(rt[0] - r)      # noqa: E701
    if pressed[8]: rt[1] = F(rt[1] + r)      # noqa: E701
    if pressed[9]: rt[1] = F(rt[1] - r)      # noqa: E701
    if pressed[10]: rt[2] = F(rt[2] + r)     # noqa: E701
    if pressed[11]: rt[2] = F(rt[2] - r)     # noqa: E701
    rot0 = [F(t) for t in rotation_deg]
    gm = rotate_vector(mv, rot0, cos, sin)
    gr = rotate_vector(rt, rot0, cos, sin)          # main.rs:111 uses the rotation before the update
    pos = [F(F(p) + g) for p, g in zip(position, gm)]
    rot = [F(q + g) for q, g in zip(rot0, gr)]
    return pos, rot, any(pressed)
