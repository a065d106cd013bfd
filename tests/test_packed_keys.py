"""CPU: the packed 4D sort keys of the quantiser (insmos_amd/csrc/common.h: pkey_make / pkey_expand, used by k_quant_keys_p) --
a host program compiled from the same header checks that the 40-bit packed key orders exactly like the canonical key4 and
expands back to it (random voxels in the packed box, the box corners, every window count used)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_keys_order_like_canonical_keys_and_expand_back(tmp_path):
    exe = tmp_path / "pkey_check"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "aux", "pkey_check.cpp"),
                           "-o", str(exe)])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120).stdout.decode()
    assert out.strip().endswith("OK"), out
