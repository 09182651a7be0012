"""Turns the four rocprofv3 --pmc passes of tools/traffic_pmc.sh into attn_traffic.json / conv_traffic.json (HBM-side bytes per launch).
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch; on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B/lane streams at 64 B,
so it is doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact."""
import csv, glob, json, os, sys, collections

out = sys.argv[1]


def per_kernel(prefix, counter):
    files = glob.glob(os.path.join(out, f"{prefix}_{counter}", "**", "*counter_collection.csv"), recursive=True)
    assert files, (prefix, counter)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def summarise(prefix, want, note, algorithmic):
    res, total = {}, 0.0
    for counter, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for k, vals in per_kernel(prefix, counter).items():
            key = next((w for w in want if w in k), None)
            if key is None:
                continue
            calls = vals[-want[key]:] if want[key] else vals        # the last launches (steady state)
            kb = sum(calls) / len(calls)
            res.setdefault(key, {})[counter + "_KB_raw"] = round(kb, 1)
            total += kb * 1024 * corr
    return dict(kernels=res, correction="gfx950: FETCH_SIZE x2 (128-B requests of 16-B/lane streams counted at 64 B); WRITE_SIZE exact",
                hbm_bytes_per_launch=int(total), algorithmic_bytes_per_launch=algorithmic, ratio=round(total / algorithmic, 2), note=note,
                source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/traffic_pmc.sh)")


attn = summarise("attn", {"mem_attn64_kernel": 3, "mem_attn_kernel": 3, "attn_combine_kernel": 3},
                 "one ppms_mem_attn call at the 1/4 scale (T=5, n=10240, 5 picked frames): attention kernel + fix-up pass + combine kernel, mean of the last 3 calls",
                 157286400)
json.dump(attn, open(os.path.join(out, "attn_traffic.json"), "w"), indent=1)
# zr1_0_x at the 1/4 scale as the loop runs it (hoisted block): activations [h | mf, hid] 384 ch in + 256 ch out as split bf16 planes (4 B per value), the
# hoisted inp share 256 ch fp32 in, weights 256 x 384 x 15 x 4 B
P = 5 * 80 * 128
conv_alg = P * (384 + 256 + 256) * 4 + 256 * 384 * 15 * 4
conv = summarise("conv", {"conv6_kernel": 3}, "one zr1_0_x launch (1,1,15), [h | mf, hid] 384 -> 256 channels + the hoisted fp32 share, "
                 "5x80x128 pixels; mean of the last 3 launches", conv_alg)
json.dump(conv, open(os.path.join(out, "conv_traffic.json"), "w"), indent=1)
# pyramid build at the 1/4 scale: both feature maps once + the five pyramid levels once (1.9375 P W values)
corr_alg = int(2 * 256 * P * 4 + 1.9375 * P * 128 * 4)
if glob.glob(os.path.join(out, "corr_FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True):
    corr = summarise("corr", {"corr_build_line_kernel": 10}, "one ppms_corr_build launch at the 1/4 scale (T=5, 256 channels, 80x128): the line-resident kernel, "
                     "mean of the last 10 of the probe's back-to-back launches (inputs resident in the Infinity Cache between launches: the fetch counter sees "
                     "what crosses the memory side)", corr_alg)
    json.dump(corr, open(os.path.join(out, "corr_traffic.json"), "w"), indent=1)
    print(json.dumps(corr)[:600])
print(json.dumps(attn)[:600])
print(json.dumps(conv)[:600])
