"""CPU-only: the host code of libvlg under AddressSanitizer (tools/asan_host_check.sh) - handle creation / argument validation /
RoPE table / error paths exercised by the C-ABI tests leave no out-of-bounds access, use-after-free or double free.  GPU-side
sanitizers are not available on this pool, so device code is compiled uninstrumented."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_code_is_asan_clean():
    if os.environ.get("VLG_LIB_PATH"):
        pytest.skip("already running under the sanitizer build")
    r = subprocess.run([os.path.join(ROOT, "tools", "asan_host_check.sh")], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "AddressSanitizer" not in out, out[-3000:]
