"""Development-only: builds profiles/rNN_pmc_gather.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> ["<bench command>"]

The file records the sha256 of csrc/kpconv.hip it was taken with: bench.py reports roofline.traffic only
while the kernel source still has that hash."""
import collections, csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KPCONV = os.path.join(ROOT, "enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd", "csrc", "kpconv.hip")
CMD = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"


def per_launch(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "kpconv" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "")
        name = name.split(">(")[0] + ">" if ">(" in name else name.split("(")[0]
        a = acc[(name, int(r["Grid_Size"]))]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
out = {"kpconv_hip_sha256": hashlib.sha256(open(KPCONV, "rb").read()).hexdigest(), "command": CMD,
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + CMD + "; per-dispatch averages; counters are KiB; traffic_bytes = "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of "
                 "16-B-per-lane coalesced reads; WRITE_SIZE is exact for 16-B-per-lane stores)",
       "launches": []}
for key in sorted(fetch, key=lambda k: -fetch[k][1]):
    if key not in write:
        continue
    f, w = fetch[key][1], write[key][1]
    out["launches"].append({"kernel": key[0], "grid_threads": key[1], "dispatches": fetch[key][0], "FETCH_SIZE_KiB": f,
                            "WRITE_SIZE_KiB": w, "traffic_bytes": (2 * f + w) * 1024})
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["launches"][:3], indent=1))
