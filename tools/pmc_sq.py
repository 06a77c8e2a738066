"""Per-kernel SQ counter ratios of the last training step of `rocprofv3 --pmc SQ_... -- python3 bench.py --graph 0`:
MFMA-busy share, wave-time split (issuing / issue-stalled / waiting), LDS bank-conflict share.
Usage: python tools/pmc_sq.py <rocprofv3 output dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
first = [int(r["Dispatch_Id"]) for r in rows if "k_pack_input" in r["Kernel_Name"]]
start = max(first)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if int(r["Dispatch_Id"]) < start:
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-44s %9s %9s | wave time: %7s %7s %7s | %9s" % ("kernel", "MFMA busy", "of peak*", "issuing", "stalled", "waiting", "LDS confl"))
tot = collections.defaultdict(float)
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    busy = c.get("SQ_BUSY_CYCLES", 0.0)
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    if busy <= 0 or wave <= 0:
        continue
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0            # rocprofv3 sums the 8 XCDs
    print("%-44s %8.1f%% %9s | %21.1f%% %6.1f%% %6.1f%% | %8.1f%%" % (
        k, 100 * mf / (gui * 256 * 4) if gui else float("nan"), "", 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wave,
        100 * c.get("SQ_WAIT_INST_ANY", 0) / wave, 100 * c.get("SQ_WAIT_ANY", 0) / wave,
        100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / lds if lds else 0))
print("* SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 256 CUs x 4 SIMDs): share of the kernel's time the MFMA pipes were occupied")
