"""Static scan of the hot kernels' gfx950 assembly (hipcc -S of csrc/*.hip into /tmp): per kernel the vector loads, the full drains
(`s_waitcnt vmcnt(0)`), the counted waits, loads that are waited for at once (a load followed within two instructions by a full
drain: serialized round trips) and scratch accesses (an array or spill in private memory: every access is a vector-memory
operation the compiler drains behind).  Round 5 found fp_fwd_rows2_kernel's stage array in scratch and head_fwd_mfma_kernel's nine
row loads issued one by one with it.     python scripts/isa_scan.py [name filter ...]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/sn2_isa"
os.makedirs(out, exist_ok=True)
for f in ("fp", "sa_mfma", "misc", "project", "loss", "sa", "geometry"):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", f"{out}/{f}.s",
                    f"{ROOT}/stratanet2_vegetation_coverage_maps_amd/csrc/{f}.hip"], check=True, stderr=subprocess.DEVNULL)
filters = sys.argv[1:]
rows = []
for f in sorted(glob.glob(out + "/*.s")):
    txt = open(f).read()
    for m in re.finditer(r"^(_Z\S+?):.*?\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
        if filters and not any(w in dem for w in filters):
            continue
        lines = [l for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        loads = [i for i, l in enumerate(lines) if re.search(r"(global|buffer)_load", l)]
        n0 = sum(1 for l in lines if "s_waitcnt vmcnt(0)" in l)
        nc = sum(1 for l in lines if re.search(r"s_waitcnt vmcnt\([1-9]", l))
        ser = sum(1 for i in loads if any("s_waitcnt vmcnt(0)" in lines[j] for j in range(i + 1, min(i + 3, len(lines)))))
        sc = sum(1 for l in lines if "scratch_" in l)
        rows.append((dem, len(lines), len(loads), n0, nc, ser, sc))
print(f"{'kernel':72s} {'instr':>6s} {'loads':>6s} {'drain':>6s} {'counted':>8s} {'load->drain':>12s} {'scratch':>8s}")
for r in sorted(rows, key=lambda r: -r[5] - r[6]):
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:6d} {r[3]:6d} {r[4]:8d} {r[5]:12d} {r[6]:8d}")
