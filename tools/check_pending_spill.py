"""Build-time guard for the persistent GEMM kernels' tile queue.  pp_q_fetch issues `global_atomic_add vN, ..., sc0` from inline
assembly and its result register is valid only behind pp_q_wait (`s_waitcnt vmcnt(0) ; tile queue: vN`); the compiler is not told
(gemm.hip: pp_q_fetch).  If register pressure makes it SPILL vN in between, the spill stores the register before the atomic has
answered and the reload hands the kernel a garbage tile index - tiles skipped or a hang, found in round 5 on a kernel variant with
30 spilled registers.  This script disassembles the device code of the built objects and fails if any kernel stores a pending
queue register to scratch.       python tools/check_pending_spill.py [object files ...]       (default: csrc/gemm.o, gemm_f16.o)"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def device_asm(obj):
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "x.o")
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "x.o"], cwd=d, capture_output=True, check=True)   # -> x.o.0.<target>
        dev = [f for f in os.listdir(d) if "amdgcn" in f]
        assert dev, "no device code in " + obj
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(d, dev[0])],
                              capture_output=True, text=True, check=True).stdout


def check(obj):
    cur, pend, bad, n_fetch = None, {}, [], 0
    for line in device_asm(obj).split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            cur, pend = m.group(1), {}
            continue
        m = re.search(r"global_atomic_add v(\d+), v\d+, v\d+, s\[\d+:\d+\] sc0", line)
        if m:
            pend[int(m.group(1))] = True
            n_fetch += 1
        # a full wait releases every pending queue register (pp_q_wait is `s_waitcnt vmcnt(0)`; the disassembly has no comments)
        if re.search(r"s_waitcnt\b.*vmcnt\(0\)", line):
            pend = {}
        m = re.search(r"scratch_store_dword(x(\d))? off, v(\[(\d+):(\d+)\]|(\d+))", line)
        if m and pend:
            regs = range(int(m.group(4)), int(m.group(5)) + 1) if m.group(4) else [int(m.group(6))]
            bad += [(cur, r) for r in regs if r in pend]
    return n_fetch, bad


def main(objs):
    total, bad = 0, []
    for o in objs:
        n, b = check(o)
        total += n
        bad += b
    if bad:
        print("FAIL: a pending tile-queue register is spilled in:", sorted(set(k for k, _ in bad)))
        return 1
    print("tile-queue registers: %d in-flight fetches checked in %d object(s), none spilled" % (total, len(objs)))
    return 0 if total else 2


if __name__ == "__main__":
    objs = sys.argv[1:] or [os.path.join(ROOT, "tiny-newsrec_amd", "csrc", f) for f in ("gemm.o", "gemm_f16.o")]
    sys.exit(main(objs))
