"""Build-time guard for the persistent GEMM kernels' tile queue.  pp_q_fetch issues `global_atomic_add vN, ..., sc0` from inline
assembly and its result register is valid only behind pp_q_wait (`s_waitcnt vmcnt(0) ; tile queue: vN`); the compiler is not told
(gemm.hip: pp_q_fetch).  Anything the compiler does WITH vN in between reads the register before the atomic has answered (or
overwrites what the atomic is about to deliver): a spill to scratch (found in round 5 on a kernel variant with 30 spilled
registers: garbage tile indices, tiles skipped or a hang), a spill to an AGPR (`v_accvgpr_write_b32 aK, vN`, what gfx90a+ tries
first), a `v_mov_b32` copy in front of the wait, a `v_readfirstlane_b32`, a phi copy on one of the paths.

This script disassembles the device code of the built objects, rebuilds every kernel's control-flow graph from the branch
targets, and runs a forward data flow over it: a queue fetch (the atomic between the two `s_mov_b64 exec` of pp_q_fetch's asm
block) makes its destination register PENDING on every path that leaves it; an executed `s_waitcnt` with `vmcnt(0)` releases
every pending register on that path; ANY other instruction that names a pending register - as a source, as a destination, alone
or inside a register range - fails the build.  Paths are followed through loops and both sides of every branch, so a use that
sits textually behind some wait but is reachable around it is found as well.
    python tools/check_pending_spill.py [object files ...]       (default: csrc/gemm.o, gemm_f16.o)
    python tools/check_pending_spill.py --selftest               (the checker against hand-written listings)"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def device_asm(obj):
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "x.o")
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "x.o"], cwd=d, capture_output=True)   # -> x.o.0.<target> (+ a host part it cannot read)
        dev = [f for f in os.listdir(d) if "amdgcn" in f]
        assert dev, "no device code in " + obj
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(d, dev[0])],
                              capture_output=True, text=True, check=True).stdout


_FUNC = re.compile(r"^([0-9a-f]+) <([\w.$]+)>:")
_INSN = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<([\w.$]+)(?:\+0x([0-9a-fA-F]+))?>\s*$")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_FETCH = re.compile(r"^global_atomic_add v(\d+), v\d+, v\d+, s\[\d+:\d+\](?: offset:\d+)? sc0")
_EXEC_MOV = re.compile(r"^s_mov_b64 exec, s\[\d+:\d+\]")


def parse(text):
    """-> {kernel: [(addr, text, branch target addr or None), ...]}"""
    funcs, cur, start = {}, None, {}
    for line in text.split("\n"):
        m = _FUNC.match(line)
        if m:
            cur = m.group(2)
            start[cur] = int(m.group(1), 16)
            funcs[cur] = []
            continue
        m = _INSN.match(line)
        if not m or cur is None:
            continue
        ins, addr = m.group(1), int(m.group(2), 16)
        tgt = None
        if ins.startswith("s_cbranch") or ins.startswith("s_branch"):
            t = _TARGET.search(line)
            assert t, "branch without a printed target: " + line
            tgt = start.get(t.group(1), None)
            assert tgt is not None, "branch into another symbol: " + line
            tgt += int(t.group(2), 16) if t.group(2) else 0
        funcs[cur].append((addr, ins, tgt))
    return funcs


def vregs(ins):
    out = set()
    for m in _VREG.finditer(ins):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, insns):
    """-> (number of queue fetches, [(address, instruction, register), ...] violations)"""
    n = len(insns)
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    fetch = {}
    for i, (a, ins, _) in enumerate(insns):
        m = _FETCH.match(ins)
        if m and 0 < i < n - 1 and _EXEC_MOV.match(insns[i - 1][1]) and _EXEC_MOV.match(insns[i + 1][1]):
            fetch[i] = int(m.group(1))
    if not fetch:
        return 0, []
    state_in = [None] * n                      # set of pending registers on entry, None = not reached with anything pending yet
    work = []
    bad = {}

    def push(i, s):
        if i >= n:
            return
        if state_in[i] is None or not s <= state_in[i]:
            state_in[i] = set(s) | (state_in[i] or set())
            work.append(i)

    for i in fetch:                            # everything before the first fetch is reached with the empty set: start at the fetches
        push(i, set())
    while work:
        i = work.pop()
        s = set(state_in[i])
        a, ins, tgt = insns[i]
        if i in fetch:
            hit = (vregs(ins) - {fetch[i]}) & s            # the fetch's own operands may not be pending either
            s.add(fetch[i])
        elif ins.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", ins):
            hit, s = set(), set()
        else:
            hit = vregs(ins) & s
        for r in hit:
            bad[(a, r)] = ins
        if ins.startswith("s_setpc") or ins.startswith("s_swappc"):
            if s:
                bad[(a, -1)] = ins + "   (indirect branch with a pending register)"
            continue
        if ins.startswith("s_endpgm"):
            continue
        if not s and i not in fetch:
            continue                                           # nothing pending on this path any more
        if tgt is not None:
            assert tgt in index, "%s: branch target %x is not an instruction" % (name, tgt)
            push(index[tgt], s)
            if ins.startswith("s_branch"):
                continue
        push(i + 1, s)
    return len(fetch), [(a, ins, r) for (a, r), ins in sorted(bad.items())]


def check(obj_or_text, is_text=False):
    text = obj_or_text if is_text else device_asm(obj_or_text)
    total, bad = 0, []
    for name, insns in parse(text).items():
        k, b = check_kernel(name, insns)
        total += k
        bad += [(name,) + x for x in b]
    return total, bad


_SELF = """
0000000000001000 <k_ok>:
	s_mov_b64 s[2:3], exec                                     // 000000001000: 00
	s_mov_b64 exec, s[4:5]                                     // 000000001004: 00
	global_atomic_add v7, v1, v2, s[8:9] sc0                   // 000000001008: 00
	s_mov_b64 exec, s[2:3]                                     // 000000001010: 00
	v_add_u32_e32 v3, v4, v5                                   // 000000001014: 00
	s_cbranch_scc1 2                                           // 000000001018: 00 <k_ok+0x24>
	v_mov_b32_e32 v9, v3                                       // 00000000101C: 00
	s_branch 65531                                             // 000000001020: 00 <k_ok+0x14>
	s_waitcnt vmcnt(0)                                         // 000000001024: 00
	v_readfirstlane_b32 s0, v7                                 // 000000001028: 00
	s_endpgm                                                   // 00000000102C: 00
0000000000002000 <k_%s>:
	s_mov_b64 s[2:3], exec                                     // 000000002000: 00
	s_mov_b64 exec, s[4:5]                                     // 000000002004: 00
	global_atomic_add v7, v1, v2, s[8:9] sc0                   // 000000002008: 00
	s_mov_b64 exec, s[2:3]                                     // 000000002010: 00
	s_cbranch_scc1 2                                           // 000000002014: 00 <k_%s+0x20>
	s_waitcnt vmcnt(0)                                         // 000000002018: 00
	s_branch 1                                                 // 00000000201C: 00 <k_%s+0x24>
	%s                                                         // 000000002020: 00
	s_waitcnt vmcnt(0)                                         // 000000002024: 00
	s_endpgm                                                   // 000000002028: 00
"""


def selftest():
    cases = [("spill", "scratch_store_dword off, v7, off offset:4"), ("agpr", "v_accvgpr_write_b32 a3, v7"), ("mov", "v_mov_b32_e32 v9, v7"),
             ("rfl", "v_readfirstlane_b32 s0, v7"), ("range", "global_store_dwordx4 v0, v[4:7], s[0:1]"), ("clobber", "v_mov_b32_e32 v7, 0"),
             ("fine", "v_mov_b32_e32 v9, v8")]
    for tag, ins in cases:
        n, bad = check(_SELF % (tag, tag, tag, ins), is_text=True)
        names = {b[0] for b in bad}
        assert n == 2 and "k_ok" not in names, (tag, n, bad)
        assert ("k_" + tag in names) == (tag != "fine"), (tag, bad)
    print("selftest: %d listings, every use of a pending register on a path around the wait is found; clean listings pass" % len(cases))
    return 0


def main(objs):
    total, bad = 0, []
    for o in objs:
        n, b = check(o)
        total += n
        bad += b
    if bad:
        print("FAIL: a pending tile-queue register is touched before its s_waitcnt vmcnt(0):")
        for name, a, ins, r in bad[:40]:
            print("   %s  +%x  v%d   %s" % (name, a, r, ins))
        return 1
    if not total:
        print("FAIL: no tile-queue fetch found in %s - has the form of pp_q_fetch's asm block (s_mov_b64 exec / global_atomic_add ... sc0 / "
              "s_mov_b64 exec) or the disassembler's operand format changed?  Update _FETCH / _EXEC_MOV in this script." % ", ".join(objs))
        return 2
    print("tile-queue registers: %d in-flight fetches followed through the control-flow graphs of %d object(s): no instruction touches a "
          "pending register before its wait" % (total, len(objs)))
    return 0


if __name__ == "__main__":
    if sys.argv[1:] == ["--selftest"]:
        sys.exit(selftest())
    objs = sys.argv[1:] or [os.path.join(ROOT, "tiny-newsrec_amd", "csrc", f) for f in ("gemm.o", "gemm_f16.o")]
    sys.exit(main(objs))
