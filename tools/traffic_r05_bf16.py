"""Summarise the FETCH_SIZE / WRITE_SIZE passes over tools/pmc_r03.py (the 16-bit kernels of the 3x3 512->512 conv @64x64,
B=32) and over tools/probes/pmc_s2d_r05.py (the stride-2 16-bit dgrad and fprop, 64 <-> 128 channels at 256x256 <-> 128x128)
into the JSON bench.py reads for the 16-bit roofline.traffic.
usage: python tools/traffic_r05_bf16.py <FETCH dir> <WRITE dir> <FETCH dir of the stride-2 run> <WRITE dir of the stride-2 run>"""
import csv, glob, json, re, sys


def per_dispatch(d, want):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(want, r["Kernel_Name"].replace(" ", "")):
                vals.append(float(r["Counter_Value"]))
    return vals


def entry(kernel_re, desc, alg, note, dirs=(1, 2)):
    fetch, write = per_dispatch(sys.argv[dirs[0]], kernel_re), per_dispatch(sys.argv[dirs[1]], kernel_re)
    assert fetch and write, "no dispatch of " + kernel_re
    f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
    hbm = int((2 * f_kb + w_kb) * 1024)
    return {"kernel": desc,
            "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/traffic_r05.sh -> tools/pmc_r03.py), mean over "
                   f"{len(fetch)} / {len(write)} dispatches; counters in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md",
            "fetch_size_kb_raw": round(f_kb, 1), "write_size_kb": round(w_kb, 1), "hbm_bytes_per_launch": hbm,
            "algorithmic_bytes_per_launch": alg, "ratio": round(hbm / alg, 3), "note": note}


x16 = 32 * 64 * 64 * 512 * 2
w16 = 9 * 512 * 512 * 2
out = {
    "direct": entry(r"conv_lp16h2_kernel<false,1,false>",
                    "conv_lp16h2_kernel<bf16, fprop> (16x16-pixel tile x 256 channels, halo resident in LDS, persistent tile lists, 16-byte "
                    "epilogue stores): 3x3 512->512 @64x64, B=32, 16-bit output - the bench.py --dtype bf16 / bf16_path roofline kernel; "
                    "plain and statistics-epilogue dispatches together",
                    2 * x16 + w16, "algorithmic = x16 134.2 MB + w16 4.7 MB read + y16 134.2 MB written"),
    "dgrad_reflect_fold": entry(r"conv_lp16h2_kernel<false,-1,true>",
                                "conv_lp16h2_kernel<bf16, dgrad, FOLD> (mmh_conv3x3_lp16 mode 2): the complete dgrad of the ReflectionPad2d(1) "
                                "3x3 512->512 conv, one fold accumulator per wave, halo offsets in LDS, persistent tile lists",
                                2 * x16 + w16, "as the fprop"),
    "wgrad": entry(r"wgrad_lp16t_kernel<false,1>",
                   "wgrad_lp16t_kernel<bf16>: 3x3 512->512 wgrad @64x64, B=32 (slab reduction not included)",
                   2 * x16 + 9 * 512 * 512 * 4, "algorithmic = x16 + dy16 268 MB read + dw 9.4 MB written; WRITE_SIZE = the split-K slabs"),
    "s2d": entry(r"conv_s2d_kernel<false>",
                 "conv_s2d_kernel<bf16> (round 5: the waves split the parity classes; weights register-resident, one dy halo per 8 x 16 "
                 "tile for all nine taps): dgrad of Conv2d(64, 128, 3, 2, 1) = forward of ConvTranspose2d(128, 64, 3, 2, 1, 1), B=32, "
                 "128x128x128 -> 256x256x64, 16-bit dx",
                 32 * 128 * 128 * 128 * 2 + 32 * 256 * 256 * 64 * 2 + 9 * 64 * 128 * 2,
                 "algorithmic = dy16 134.2 MB + w16 0.15 MB read + dx16 268.4 MB written (the 9 x 17 halos of the 8 x 16 tiles overlap: "
                 "160 MB of dy are requested)", dirs=(3, 4)),
    "wgrad_s2": entry(r"wgrad_lp16t_kernel<false,2>",
                      "wgrad_lp16t_kernel<bf16, stride 2> (round 5: nine taps of a 64 x 128 tile resident, 2 x 16-pixel blocks, 5 x 33 halo "
                      "with permuted pixel columns): wgrad of Conv2d(64, 128, 3, 2, 1), B=32, 256x256x64 / 128x128x128 (slab reduction not "
                      "included)",
                      32 * 256 * 256 * 64 * 2 + 32 * 128 * 128 * 128 * 2 + 9 * 64 * 128 * 4,
                      "algorithmic = x16 268.4 MB + dy16 134.2 MB read + dw 0.3 MB written; WRITE_SIZE = 256 split-K slabs of 295 KB; a "
                      "5-row halo per 4 input rows is requested", dirs=(3, 4)),
    "s2f": entry(r"conv_s2f_kernel<false,1,2,2,false,2,128>",
                 "conv_s2f_kernel<bf16, C = 64> (weights register-resident, de-interleaved 17 x 33 input halo per 8 x 16 output tile): "
                 "fprop of Conv2d(64, 128, 3, 2, 1), B=32, 256x256x64 -> 128x128x128, 16-bit y",
                 32 * 256 * 256 * 64 * 2 + 32 * 128 * 128 * 128 * 2 + 9 * 64 * 128 * 2,
                 "algorithmic = x16 268.4 MB + w16 read + y16 134.2 MB written", dirs=(3, 4)),
}
print(json.dumps(out, indent=2))
