"""Development-only: profiles/rNN_pmc_mfma.json from a rocprofv3 --pmc pass over tools/mfma_probe.py.
usage: pmc_mfma.py counter_collection.csv out.json"""
import collections, csv, json, sys
shapes = [(19464, 64, 990), (19464, 32, 480), (3986, 64, 960), (923, 128, 1920), (225, 256, 3840), (65, 512, 7680)]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_f32_mfma" in r["Kernel_Name"]]
by_disp = collections.OrderedDict()
for r in rows:
    d = by_disp.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"][:60], "grid": r["Grid_Size"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = list(by_disp.values())
per = len(disp) // len(shapes)
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 tools/mfma_probe.py",
       "units": "per launch (mean over the launches of a shape); GRBM_GUI_ACTIVE is summed over the 8 XCDs; mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
       "launches": []}
for i, (M, N, K) in enumerate(shapes):
    grp = disp[i * per:(i + 1) * per]
    if not grp:
        continue
    e = {"M": M, "N": N, "K": K, "kernel": grp[0]["kernel"], "grid": grp[0]["grid"]}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
        e[c] = sum(g.get(c, 0.0) for g in grp) / len(grp)
    e["mfma_utilisation"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0) if e["GRBM_GUI_ACTIVE"] else None
    e["flops_algorithmic"] = 2.0 * M * N * K
    out["launches"].append(e)
json.dump(out, open(sys.argv[2], "w"), indent=1)
for e in out["launches"]:
    print(e["M"], e["N"], e["K"], "mfma_utilisation %.3f" % e["mfma_utilisation"])
