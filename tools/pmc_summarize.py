"""Summarise rocprofv3 --pmc passes (tools/run_pmc.sh) per kernel: mean counter value per launch + derived figures.

    python tools/pmc_summarize.py gpurun_out/pmc_folded profiles/r01_pmc_summary.json
"""
import csv, glob, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous namespace" not in name or "at::native" in name:
            continue
        short = name.split("::")[1].split("(")[0]
        short = short.split("<")[0]
        d = acc.setdefault(short, {}).setdefault(r["Counter_Name"], [])
        d.append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    e = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
    der = {}
    if "FETCH_SIZE" in e:
        der["hbm_read_bytes_x2_rule"] = 2 * 1024 * e["FETCH_SIZE"]["mean"]   # gfx950: FETCH_SIZE counts half of a wide read stream
    if "WRITE_SIZE" in e:
        der["hbm_write_bytes"] = 1024 * e["WRITE_SIZE"]["mean"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        der["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (e["GRBM_GUI_ACTIVE"]["mean"] / 8 * 1024)
    e["derived"] = der
    out[k] = e
existing = {}
if os.path.exists(dst):
    existing = json.load(open(dst))
existing.update(out)
json.dump(existing, open(dst, "w"), indent=1, sort_keys=True)
print("kernels:", sorted(out))
