"""Summarise the two PMC passes of tools/traffic_r04.sh into the JSON bench.py reads for roofline.traffic.
usage: python tools/traffic_r04.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import csv, glob, json, re, sys


def per_dispatch(d, want):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(want, r["Kernel_Name"].replace(" ", "")):
                vals.append(float(r["Counter_Value"]))
    return vals


KERNEL = r"wino_gemm_kernel<128,2>"
fetch, write = per_dispatch(sys.argv[1], KERNEL), per_dispatch(sys.argv[2], KERNEL)
assert fetch and write, "no dispatch of wino_gemm_kernel<128,2> in the counter files"
f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
alg = 64 * (3872 * 512 + 512 * 512 + 3872 * 512) * 4
hbm = int((2 * f_kb + w_kb) * 1024)
print(json.dumps({"winograd_gemm": {
    "kernel": "wino_gemm_kernel<128,2> (two-level accumulation): 64 x [3872x512].[512x512], the Winograd F(6x6,3x3)-domain GEMMs of "
              "the 3x3 512->512 fprop @64x64, B=32 - the bench.py roofline kernel; the one-level dgrad twin <128,1> is NOT mixed in",
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/traffic_r04.sh -> tools/pmc_r02.py, PMC_WHICH=f32), "
           f"mean over {len(fetch)} / {len(write)} dispatches; counters are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
           "128-B requests at 64 B)",
    "fetch_size_kb_raw": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
    "ratio": round(hbm / alg, 3),
    "note": "algorithmic = V 507.5 MB + U 67.1 MB read + M 507.5 MB written"}}, indent=2))
