"""CPU: the counted-wait kernels keep their waits after compilation (tools/check_counted_waits.py).  Two guards on the BUILT
library: a performance one (a compiler that adds `vmcnt(0)` waits only slows the kernels down) and a CORRECTNESS one -- no instruction may read
the destination of one of the kernels' untracked inline-asm register loads before a `vmcnt` wait has retired that load (all vector-memory
operations are tracked in issue order; `vmcnt(N)` keeps the N youngest pending), which a compiler-inserted copy or spill of such a register
would violate."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs ROCm's llvm-objdump")
def test_tile_loops_hold_only_the_counted_waits():
    lib = os.path.join(ROOT, "semanticlidarunc_amd", "libslu_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libslu_hip.so not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_counted_waits.py"), lib], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    names = [ln.split()[0] for ln in r.stdout.splitlines() if "_h8_kernel" in ln]
    assert sum("tail2_h8_kernel" in n for n in names) == 5 and sum("ring3_h8_kernel" in n for n in names) == 9, names
