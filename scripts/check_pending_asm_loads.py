"""ISA check for the hand-waited inline-asm loads of pgemm's BatchNorm-backward epilogue (csrc/pgemm.hip): between a group of `global_load_*` statements and
the `s_waitcnt vmcnt(0)` statement that covers them, no VALU instruction may read one of the pending destination registers (the register allocator is free
to copy an asm output the moment its statement ends -- it did, with 24 registers pending: wrong values that changed from run to run).
Usage (CPU, a few seconds):  python scripts/check_pending_asm_loads.py        exits 1 when a read is found."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "pg.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{ROOT}/simpledepthestimation_amd/csrc", "-S",
                    "--cuda-device-only", "-o", out, f"{ROOT}/simpledepthestimation_amd/csrc/pgemm.hip"], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
bad = groups = 0
for f in re.split(r"\n(?=_Z\w+:)", txt):
    name = f.split(":")[0]
    if "pgemm_kernel" not in name or "Lb1E" not in name:
        continue
    lines = f.split("\n")
    i = 0
    while i < len(lines):
        if "global_load_dwordx2" in lines[i] and ";;#ASMEND" in lines[i + 1]:          # a load statement of its own: the wait is a later statement
            j, dests, reads = i, set(), []
            while j < len(lines) and not ("s_waitcnt vmcnt(0)" in lines[j] and ";;#ASMSTART" in lines[j - 1]):
                m = re.match(r"\s*global_load_dword(x2)?\s+(v\[?[\d:]+\]?)", lines[j])
                if m:
                    n = [int(x) for x in re.findall(r"\d+", m.group(2))]
                    dests.update(range(n[0], n[-1] + 1))
                elif re.match(r"\s*v_", lines[j]) and "," in lines[j]:
                    ops = lines[j].split(",", 1)[1]
                    srcs = [int(x) for x in re.findall(r"v(\d+)", ops)]
                    for a, b in re.findall(r"v\[(\d+):(\d+)\]", ops):
                        srcs += list(range(int(a), int(b) + 1))
                    if any(x in dests for x in srcs):
                        reads.append(lines[j].strip())
                j += 1
            groups += 1
            if reads:
                bad += 1
                print(f"{name[:90]}: {len(reads)} VALU reads of pending load destinations before the wait, e.g. {reads[:3]}")
            i = j
        i += 1
print(f"{groups} separately-waited load groups checked, {bad} with reads of pending registers")
sys.exit(1 if bad or not groups else 0)
