#!/usr/bin/env python
"""Guard for the counted-wait kernels (tail2_h8_kernel, ring3_h8_kernel: DESIGN 3.1g).  Their speed -- not their results -- depends on two
things the compiler is free to change: that it leaves the `s_waitcnt vmcnt(N)` of the source alone, and that it does not add its own
`vmcnt(0)` inside the tile loop (it does for LDS reads without a TBAA tag and for tracked global loads once LDS-DMAs are interleaved).  This
script disassembles the gfx950 code objects of libslu_hip.so and reports, per instantiation, the vmcnt values in the tile loop; it exits
non-zero when a loop contains more `vmcnt(0)` than the first-tile branches account for, or when an instruction reads the destination of
one of the kernels' untracked (inline-asm) register loads before a vmcnt wait has RETIRED that load (that one WOULD be a wrong result): all
vector-memory operations are tracked in issue order and `vmcnt(N)` retires all but the N youngest, so a compiler-inserted move or spill of such
a register behind a counted wait with N > 0 is caught too.  The scan is linear over the disassembly (not branch-aware).

    python tools/check_counted_waits.py [path/to/libslu_hip.so]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                         "semanticlidarunc_amd", "libslu_hip.so"))
KERNELS = re.compile(r"^(\S*(tail2_h8_kernel|ring3_h8_kernel)\S*)>?:$")
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    local = os.path.join(tmp, "lib.so")
    shutil.copy(so, local)                                  # llvm-objdump --offloading writes the bundles next to its input
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
        name, waits, nbar = None, [], 0
        vm_ops = []        # vector-memory operations issued so far in program order: the destination registers of an untracked register
                           # load, or an empty set for anything else that counts in vmcnt (stores, LDS-DMA, tracked loads, atomics)
        def pending_regs():
            r = set()
            for regs in vm_ops:
                r |= regs
            return r
        def flush():
            global bad
            if name is None:
                return
            inner = [w for w in waits]
            zeros = sum(1 for w in inner if w == 0)
            counted = [w for w in inner if w > 0]
            # expected vmcnt(0): set-up (<= 6), one per first-tile position (<= positions), the kernel's exit, RESWAIT == 0 forms (<= 1)
            limit = 6 + max(1, nbar) + 2
            flag = "" if zeros <= limit and counted else "   <-- more vmcnt(0) than the first-tile branches explain" if counted else "   <-- no counted wait left"
            if flag:
                bad += 1
            print(f"{name[:90]:90s} barriers {nbar:2d}  counted {sorted(set(counted))}  vmcnt(0) x{zeros}{flag}")
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                flush()
                name = m.group(1) if ("tail2_h8_kernel" in m.group(1) or "ring3_h8_kernel" in m.group(1)) else None
                waits, nbar = [], 0
                vm_ops.clear()
                continue
            if name is None:
                continue
            ins = line.split("//")[0]
            w = re.search(r"s_waitcnt\s+vmcnt\((\d+)\)", ins)
            if w:
                n = int(w.group(1))
                waits.append(n)
                # vmcnt(N) retires everything but the N YOUNGEST vector-memory operations: their destinations stay pending
                del vm_ops[:max(0, len(vm_ops) - n)]
            if "s_barrier" in ins:
                nbar += 1
            # the untracked register loads (inline asm, SGPR base + VGPR offset): nothing may read their destination before a wait retires them
            m2 = re.search(r"global_load_dwordx[24]\s+v\[(\d+):(\d+)\],\s*v\d+,\s*s\[", ins)
            if m2:
                vm_ops.append(set(range(int(m2.group(1)), int(m2.group(2)) + 1)))
                continue
            if re.search(r"\b(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|scratch_load|scratch_store)", ins):
                vm_ops.append(set())                       # counts in vmcnt, holds no untracked destination
                continue
            pend = pending_regs()
            if pend:
                ops = ins.strip().split(None, 1)
                srcs = ops[1].split(",", 1)[1] if len(ops) > 1 and "," in ops[1] else ""
                if ops and ops[0].startswith(("ds_write", "v_mfma", "s_")):
                    srcs = ops[1] if len(ops) > 1 else ""       # no destination operand first / accumulate in place
                used = set()
                for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", srcs):
                    used.update(range(int(a), int(b) + 1) if a else [int(c)])
                if used & pend:
                    print(f"{name[:90]}: {ins.strip()[:80]}   <-- reads a register of an untracked load that no vmcnt wait has retired yet")
                    bad += 1
                    for regs in vm_ops:
                        regs -= used
        flush()
sys.exit(1 if bad else 0)
