"""For every kernel of a translation unit: the s_waitcnt vmcnt(N) / lgkmcnt(N) instructions, scratch accesses and barriers inside
its innermost loops (a full vmcnt(0) drain inside a k-loop that keeps LDS-DMA in flight is how conv_s2d_kernel lost a third of
its time).   python tools/isa_waits.py mmhand_amd/csrc/<file>.hip [kernel-name-substring]"""
import re, subprocess, sys, os
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/_waits.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-Wno-unused-function",
                "-Wno-inline-asm", "-S", "--cuda-device-only", "-o", out, os.path.basename(src)], cwd=os.path.dirname(os.path.abspath(src)),
               stderr=subprocess.DEVNULL, check=True)
lines = open(out).read().split("\n")
kern, start = None, 0
kernels = []
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m: kern, start = m.group(1), i
    if kern and l.strip().startswith(".amdhsa_kernel " + kern):
        kernels.append((kern, start, i)); kern = None
for name, a, b in kernels:
    if pat not in name: continue
    body = lines[a:b]
    # innermost loops: label of "Inner Loop Header" .. the last branch back to it
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        if "Inner Loop Header" in l or ("Loop Header" in l and "Depth=" in l):
            # the label is on the previous non-comment line
            j = i               # the label sits on this line (".LBB0_3: ; =>This Loop Header") or on the one above
            while j >= 0 and not re.match(r"^(\.LBB\d+_\d+):", body[j]): j -= 1
            if j < 0: continue
            lab = re.match(r"^(\.LBB\d+_\d+):", body[j]).group(1)
            ends = [k for k, x in enumerate(body) if re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b", x) and k > j]
            if ends: loops.append((lab, j, max(ends), "Inner" in l))
    sz = next((x.split()[-1] for x in body if ".amdhsa_private_segment_fixed_size" in x), "?") if False else "?"
    scratch = sum("scratch_" in x for x in body)
    print(f"{name[:90]}: scratch ops {scratch}, loops {len(loops)}")
    for lab, s0, s1, inner in loops:
        seg = body[s0:s1 + 1]
        mf = sum("v_mfma" in x for x in seg)
        if mf == 0: continue
        w = [x.strip().split(";")[0].strip() for x in seg if "s_waitcnt" in x]
        vm0 = sum(1 for x in w if re.search(r"vmcnt\(0\)", x))
        print(f"    loop {lab} ({s1 - s0} lines, {mf} MFMAs, {sum('global_load_lds' in x for x in seg)} DMA, {sum('ds_read' in x for x in seg)} ds_read, "
              f"{sum('scratch_' in x for x in seg)} scratch, {sum('s_barrier' in x for x in seg)} barriers): vmcnt(0) x{vm0}; waits: {', '.join(sorted(set(w)))[:300]}")
