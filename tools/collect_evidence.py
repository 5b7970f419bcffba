#!/usr/bin/env python
"""Assemble profiles/r<NN>_* from the scratch directory of one evidence pass on the GPU box (see profiles/README.md for the command
that fills it): bench line, rocprofv3 summaries (list executor + eager timeline), per-kernel PMC table and the HBM-traffic JSON that
bench.py reports as roofline.traffic.   python tools/collect_evidence.py gpurun_out/<dir> gpurun_out/<tag>_step r03"""
import json
import os
import re
import shutil
import sys

src, step, rnd = sys.argv[1].rstrip("/") + "/", sys.argv[2], sys.argv[3]
P = lambda n: os.path.join("profiles", f"{rnd}_{n}")


def last_json(path):
    return [l for l in open(path).read().splitlines() if l.startswith("{")][-1]


open(P("bench_n1.json"), "w").write(last_json(src + "bench_n1.json") + "\n")
open(P("bench_profiled.json"), "w").write(last_json(src + "bench_profiled.json") + "\n")
if os.path.exists(src + "other_configs.jsonl"):      # (round 3: a separate tools/bench_configs.py run; since round 4 the bench line carries them)
    oc = [l for l in open(src + "other_configs.jsonl").read().splitlines() if l.startswith("{")]
    open(P("other_configs.jsonl"), "w").write("\n".join(oc) + "\n")
shutil.copy(src + "pmc_per_kernel.txt", P("pmc_bench_per_kernel.txt"))
shutil.copy(src + "gpu_tests.txt", P("gpu_tests.txt"))
shutil.copy(src + "bench_list_kernel_summary.txt", P("bench_kernel_summary.txt"))
shutil.copy(step + "_kernel_summary.txt", P("bench_kernel_summary_eager.txt"))
shutil.copy(step + "_timeline.txt", P("step_timeline.txt"))

rows = {}
for l in open(src + "pmc_per_kernel.txt"):
    m = re.match(r"(.*\]) +(\d+) +([\d.]+) +(.*)$", l.rstrip())
    if m:
        c = dict(kv.split("=") for kv in m.group(4).split())
        rows[m.group(1).strip()] = {"calls": int(m.group(2)), "avg_us": float(m.group(3)), **{k: float(v) for k, v in c.items()}}
kb = lambda x: int(round(x * 1024))
prev = os.path.join("profiles", "r%02d_roofline_traffic.json" % (int(rnd[1:]) - 1))
old = json.load(open(P("roofline_traffic.json") if os.path.exists(P("roofline_traffic.json")) else prev))   # kernel names / notes carried over


def ent(key, name, alg):
    r = rows[key]
    f, w = kb(r["FETCH_SIZE"] * 2), kb(r["WRITE_SIZE"])
    return {"kernel": key.split(" [")[0].strip(), "what": old.get(name, {}).get("what", old.get(name, {}).get("kernel", name)), "fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w, "algorithmic_min_bytes": alg,
            "avg_us_profiled": r["avg_us"], "l2_hit_rate": round(r["TCC_HIT"] / (r["TCC_HIT"] + r["TCC_MISS"]), 3),
            "mfma_busy_cycles": r["SQ_VALU_MFMA_BUSY_CYCLES"], "gui_active_cycles": r["GRBM_GUI_ACTIVE"]}


def find(*prefixes):
    """The row of the first prefix that occurs (a tag's kernel changes from round to round), the longest-running one of that name."""
    for prefix in prefixes:
        ks = [k for k in rows if k.startswith(prefix)]
        if ks:
            return max(ks, key=lambda k: rows[k]["avg_us"] * rows[k]["calls"])
    raise KeyError(prefixes)


def opt(name, alg, *prefixes):
    try:
        return {name: ent(find(*prefixes), name, alg)}
    except KeyError:
        return {}


red = rows[find("splitk_reduce_kernel<bf16>")]
dx = ent(find("g32_kernel<0, 0, float", "gemm256_kernel<0, 0, float>", "gemm256_kernel<0, 1, float>"), "gen_dx", 333000000)
dx["reduce_fetch_bytes"], dx["reduce_write_bytes"] = kb(red["FETCH_SIZE"] * 2), kb(red["WRITE_SIZE"])
dx["hbm_bytes"] += dx["reduce_fetch_bytes"] + dx["reduce_write_bytes"]
ls = rows[find("sce_loss_kernel<bf16")]
out = {"_source": f"FINAL code of round {int(rnd[1:])}: rocprofv3 --pmc passes of bench.py (tools/pmc_bench.sh -> profiles/{rnd}_pmc_bench_per_kernel.txt), per launch; "
       "fetch_bytes = FETCH_SIZE [KB] x 1024 x 2 (gfx950: the counter counts 64-byte units as 32), write_bytes = WRITE_SIZE [KB] x 1024; "
       f"calibrated in the same run on sce_loss_kernel: 2 x {ls['FETCH_SIZE']:.4g} KB = {ls['FETCH_SIZE'] * 2 * 1024 / 1e6:.1f} MB fetched / "
       f"{ls['WRITE_SIZE']:.4g} KB = {ls['WRITE_SIZE'] * 1024 / 1e6:.1f} MB written against the 297.0 MB of bf16 logits it reads and the 297.0 MB gradient it writes",
       "gen_fwd": ent(find("gemm256_kernel<0, 1, bf16>"), "gen_fwd", 333000000), "gen_dx": dx,
       # the vocabulary weight gradient WITH the optimizer epilogue: dlogits 297.0 MB + y 5.0 MB read, then per parameter (15,627,264)
       # p / m / v read and written + the bf16 shadow written = 26 B (no gradient store); without the epilogue it was 364.5 MB
       "gen_dw": ent(find("g32_kernel<1, 0, float", "gemm_bf16_v2_kernel<float, 1, 0, 128, 128, 2, 2, 4>"), "gen_dw", 297000000 + 5000000 + 26 * 15627264),
       "sce_loss": ent(find("sce_loss_kernel<bf16"), "sce_loss", 593952768), "adam": ent(find("adam_ranges_kernel", "adam_kernel"), "adam", 0),
       **opt("adam2d", 0, "adam2d_kernel"), "embed_bwd": ent(find("embed_bwd_kernel<bf16>"), "embed_bwd", 4864 * (1024 + 2048)),
       # the sample-stationary stacks: every workgroup streams all of the stack's weights through its XCD's L2 (algorithmic = the
       # weights once + the features / ids read + the tensors saved for the backward)
       "enc_stack_fwd": ent(find("layer_ss_fwd_kernel<false"), "encoder stack forward (2 layers + front end), one launch", 0),
       "dec_stack_fwd": ent(find("layer_ss_fwd_kernel<true"), "decoder stack forward (2 layers + embedding), one launch", 0)}
# bench.py looks a bracket's traffic up by its tag
out["loss"], out["ss_enc"], out["ss_dec"] = out["sce_loss"], out["enc_stack_fwd"], out["dec_stack_fwd"]
json.dump(out, open(P("roofline_traffic.json"), "w"), indent=1)
d = json.loads(last_json(src + "bench_n1.json"))
print(f"{d['value']:.0f} samples/s, {d['ms_per_step']} ms/step; roofline {d['roofline']['kernel_tag']} {d['roofline']['frac']}; "
      f"north_star {d['north_star']['ms']} ms; decode {d['decode']['batch1']['us_per_step_replay_only']} / "
      f"{d['decode']['batch128']['us_per_step_replay_only']} us")
for k in ("gen_fwd", "gen_dx", "gen_dw", "sce_loss"):
    print(k, round(out[k]["hbm_bytes"] / 1e6, 1), "MB HBM-side,", round(out[k]["hbm_bytes"] / max(out[k]["algorithmic_min_bytes"], 1), 2), "x algorithmic")
