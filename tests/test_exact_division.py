"""pt_div_by_reciprocal (csrc/pt_exact_div.h) must equal the IEEE quotient bit for bit on its
whole admitted domain: ray_box uses it in place of the shader's divisions (shader.wgsl:241-242).
CPU only: compiles a small C harness with gcc and runs ~3e8 random + adversarial cases."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reciprocal_division_is_bit_exact(tmp_path):
    exe = str(tmp_path / "div_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "div_check.c"), "-lm"], check=True)
    out = subprocess.run([exe, "100000000"], check=True, capture_output=True, text=True, timeout=600).stdout.split()
    total, bad = int(out[0]), int(out[1])
    assert total > 250_000_000
    assert bad == 0
