#!/usr/bin/env python3
"""Build guard against a code-generation fault of this toolchain (ROCm 7.2 clang, AMDGPU backend).

Round 3's memory access fault in ``k_step_wave2<kMixture, true>`` ("The +Acc fault")
was this: the register allocator split the live range of the guide index ``g`` around the Stirling shift
loop of ``lgamma_digamma_diff2`` and put the saving copy ``v_mov_b32 v124, v116`` at the HEAD OF THE JOIN
BLOCK of the ``if`` that guards the loop - in front of the ``s_or_b64 exec, exec, s[0:1]`` that re-enables
the lanes which skipped the ``if``::

    s_and_saveexec_b64 s[0:1], s[4:5]      ; lanes whose argument needs shifting
    s_cbranch_execz  .LBB39_150
    ...                                    ; the loop
    s_or_b64 exec, exec, s[4:5]
  .LBB39_150:
    v_mov_b32_e32 v124, v116               ; <- runs for the lanes of the `then` side only
    s_or_b64 exec, exec, s[0:1]            ; <- the other lanes come back HERE
    ...  v116 reused  ...  v_mov_b32_e32 v116, v124   ; every other lane now holds garbage in g

The lanes that did not take the ``if`` never copy, ``v116`` is then reused as a temporary, and the later
restore hands them garbage: ``g`` is an address index afterwards.  Nothing in the source is wrong; where
the allocator splits moves with register pressure (a one-instruction change in ``frcp`` produced it), so
no kernel is safe by inspection of its source.  Every build is therefore scanned:

    python scripts/check_exec_join.py crispr-bean_amd/lib/libbean_hip.so [...]

For every ``s_and_saveexec_b64 sX, ... ; s_cbranch_execz L`` the instructions from ``L`` up to the first write
of ``exec`` are looked at; if that write is ``s_or_b64 exec, exec, sX`` (the join of that ``if``) and a vector instruction (lane-crossing
``v_readlane`` / ``v_writelane`` excepted: they ignore exec) stands in front of it, that is a finding.
``_lib.build_library`` fails the build on any finding; ``tests/test_isa_check.py`` keeps the committed
build honest and pins the detector on a minimal code object with the pattern.
"""
import re
import subprocess
import sys
import tempfile
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"

EXEC_RESTORE = re.compile(r"^(s_or_b64\s+exec,|s_or_saveexec_b64|s_andn2_saveexec_b64|s_and_saveexec_b64|s_mov_b64\s+exec,|"
                          r"s_xor_b64\s+exec,|s_andn2_b64\s+exec,|s_and_b64\s+exec,)")
VECTOR = re.compile(r"^(?!v_writelane|v_readlane|v_readfirstlane)(v_|ds_|global_|scratch_|buffer_|flat_)")
# a VGPR-to-VGPR move or a spill / reload: what live-range splitting inserts
COPY_VV = re.compile(r"^(v_mov_b32_e32\s+v\d+,\s*v\d+|v_mov_b64_e32\s+v\[\d+:\d+\],\s*v\[\d+:\d+\]|v_accvgpr_|scratch_store|scratch_load)")


def code_object(path):
    """The gfx950 code object inside a HIP shared library (its .hip_fatbin section), or the file itself
    if it already is a code object."""
    with open(path, "rb") as f:
        head = f.read(20)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":  # e_machine EM_AMDGPU
        return path, False
    sec = tempfile.NamedTemporaryFile(suffix=".fatbin", delete=False).name
    out = tempfile.NamedTemporaryFile(suffix=".co", delete=False).name
    try:
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", f".hip_fatbin={sec}", path,
                        sec + ".x"], check=True)
        subprocess.run([BUNDLER, "--type=o", "--unbundle", f"--input={sec}", f"--output={out}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    finally:
        for f in (sec, sec + ".x"):
            if os.path.exists(f):
                os.unlink(f)
    return out, True


def scan(path):
    co, tmp = code_object(path)
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    if tmp:
        os.unlink(co)
    findings = []
    func, start = None, 0
    insns = {}   # address -> text, per function
    targets = {}  # address of a branch -> offset of its target in the function
    order = []
    funcs = []

    def flush():
        if func is not None:
            funcs.append((func, start, dict(insns), list(order), dict(targets)))

    for line in dis.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            flush()
            func, start = m.group(2), int(m.group(1), 16)
            insns.clear()
            order.clear()
            targets.clear()
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and func is not None:
            a = int(m.group(2), 16)
            insns[a] = m.group(1)
            tm = re.search(r"<[^<>]*\+0x([0-9a-f]+)>\s*$", line)
            if tm:
                targets[a] = int(tm.group(1), 16)
            order.append(a)
    flush()
    for name, st, ins, od, tg in funcs:
        idx = {a: i for i, a in enumerate(od)}
        seen = set()
        for i, a in enumerate(od):
            t = ins[a]
            if not t.startswith("s_cbranch_execz") or a not in tg or i == 0:
                continue
            # the idiom: s_and_saveexec_b64 sX, cond (or the else form) ; s_cbranch_execz L  ...  L: s_or_b64 exec, exec, sX
            # - whatever vector instruction sits between L and that s_or_b64 runs for the lanes that took the branch
            #   body only (for none at all when the s_cbranch_execz itself was taken)
            tgt = st + tg[a]
            if tgt not in idx or tgt in seen:
                continue
            # the mask this branch's `if` saved: the s_and_saveexec_b64 (else form: s_xor_b64 exec, exec, sX) right
            # in front of it.  A branch without one is not the head of an `if`: the compiler also emits
            # `s_cbranch_execz 1 ; s_branch n` hops INSIDE a divergent region, whose target is ordinary region code
            # in front of the region's own restore.
            saved = None
            for k in range(i - 1, max(i - 4, -1), -1):
                m = re.match(r"(?:s_and_saveexec_b64|s_andn2_saveexec_b64)\s+(s\[\d+:\d+\]|vcc)\s*,", ins[od[k]]) or \
                    re.match(r"s_xor_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)\s*$", ins[od[k]])
                if m:
                    saved = m.group(1)
                    break
                if not ins[od[k]].startswith("s_") or EXEC_RESTORE.match(ins[od[k]]):
                    break
            if saved is None:
                continue
            seen.add(tgt)
            pre, restored = [], False
            for j in range(idx[tgt], min(idx[tgt] + 64, len(od))):
                tx = ins[od[j]]
                m = re.match(r"s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)\s*$", tx)
                if m:
                    restored = m.group(1) == saved
                    break
                if EXEC_RESTORE.match(tx) or tx.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                    break
                if VECTOR.match(tx):
                    pre.append((od[j], tx))
            if pre and restored:
                kind = "copy" if any(COPY_VV.match(tx) for _, tx in pre) else "other"
                findings.append((name, tgt, kind, pre))
    return findings


def findings_of(path):
    """[(function, address, kind, [(address, instruction)])] of one library / code object"""
    return scan(path)


def main(argv):
    strict = "--copies-only" not in argv
    paths = [a for a in argv if not a.startswith("--")]
    bad = 0
    for p in paths:
        fs = scan(p)
        n_copy = sum(1 for f in fs if f[2] == "copy")
        print(f"{p}: {len(fs)} join blocks with vector instructions in front of the exec restore ({n_copy} with copies)")
        for name, tgt, kind, pre in fs:
            if kind == "copy" or strict or "--all" in argv:
                print(f"  [{kind}] {name} @ {tgt:#x}")
                for a, tx in pre:
                    print(f"      {a:#x}: {tx}")
        bad += n_copy if not strict else len(fs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
