"""bench.py — headline benchmark of the Factorizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): volumes/sec, forward+backward(+AdamW) of the README Swin Factorizer
(in=4, out=3, 128^3, widths 32-512, head_dim 8, patch 8, HALS rank 1 x 5 iterations), per-GPU
batch 2, data-parallel over N GPUs with one flat-bucket RCCL all-reduce per step
(BASELINE.json configs[3]; weak scaling).  Synthetic inputs, random-init weights, fp32.

One JSON line is printed by rank 0.  `roofline` describes the dominant native kernel of the
timed region (HIP events on the launch stream, algorithmic bytes of SURVEY.md §8d);
`cpu_baseline` is the CPU oracle (oracle/cpu_ref.py, a port of the reference's CPU path) timed
on this host on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402
from factorizer_amd.parallel import FlatGradSync  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak, /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32, 256 CUs)
BF16_MFMA_PEAK_TFLOPS = 16 * FP32_MFMA_PEAK_TFLOPS   # same guide: the fp32 MFMA runs at 1/16 of the dense bf16 rate (~2.5 PFLOP/s)


def split_products(name, dtype):
    """bf16 MFMA products one fp32 product of this launch family is formed from (0 = it runs on v_mfma_f32_32x32x2_f32).
    fp32 storage: both operands split in three bf16 levels, the six products of weight >= 2^-24 kept (csrc/gemm_bx.h; error
    <= the fp32 MFMA's own, profiles/r03_bx6_accuracy.json).  bf16 storage: weights three levels, the column operand one
    exact term (three behind a LayerNorm / GELU prologue, which produces new fp32 values): three or six products — three
    is used for the roof (the optimistic one)."""
    if os.environ.get("FZ_GEMM_BX", "1") == "0":
        return 0
    m = re.search(r"_(\d+)(?:->|x)", name)
    k = int(m.group(1)) if m else 0
    if name.startswith(("wgrad_", "conv_k3", "mlp_chain_bwd_wgrad_", "dgrad_lnbwd_64", "conv_k2s2", "tconv_k2s2", "outproj_mlp_chain_fwd_",
                        "mlp_chain_fwd_32", "upcat_")):
        bx = True
    elif name.startswith(("ln_linear_", "act_linear_res_", "cat_linear_", "linear_dgrad_", "linear_")):
        bx = k >= 64
    else:
        bx = dtype == "bf16" and name.startswith(("mlp_chain_fwd_64", "mlp_chain_bwd_64"))
    return 0 if not bx else (6 if dtype == "f32" else 3)


def mfma_roof_tflops(name, dtype):
    """matrix-pipe roof of one launch family in fp32-equivalent TFLOP/s: 157.3 on the fp32 MFMA, dense bf16 peak / products
    per fp32 product on the split-bf16 path (419 TFLOP/s at six products)"""
    n = split_products(name, dtype)
    return FP32_MFMA_PEAK_TFLOPS if n == 0 else BF16_MFMA_PEAK_TFLOPS / n


# fp32 MFMA flop per ALGORITHMIC byte of the fused kernels whose arithmetic intensity is above the fp32 ridge
# (157.3 TFLOP/s / 8 TB/s = 19.7 flop/B).  mlp_chain_bwd_wgrad (C = 32, hidden 64): per voxel 2 input-gradient GEMMs
# + 2 weight-gradient GEMMs of 2*32*64 flop each = 16 384 flop over 5 planes of 32 channels * 4 B = 640 B.
MFMA_FLOP_PER_BYTE = {"mlp_chain_bwd_wgrad_": 16384 / 640}

# HBM traffic per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# passes of this same command, FETCH_SIZE doubled per the gfx950 correction of
# MI355X_MICROARCH.md §HBM; tools/pmc_traffic.py → profiles/rNN_pmc_traffic.json).  Counters cannot be
# read from inside the benchmark: the committed summary of the profiled run is quoted, and
# `traffic_source` says which file (with its content hash and the commit it was measured at), so a stale
# number is visible as such.
PMC_TRAFFIC = os.path.join(ROOT, "profiles", os.environ.get("FZ_PMC_TRAFFIC", "r06_pmc_traffic.json"))
PMC_TRAFFIC_BF16 = os.path.join(ROOT, "profiles", os.environ.get("FZ_PMC_TRAFFIC_BF16", "r06_pmc_traffic_bf16.json"))   # the --dtype bf16 command
# What a hand-written streaming kernel of the same read : write mix reaches on MI355X (tools/probes/mem_ceilings.hip, 16 B per
# lane, best over 4 / 8 / 16 waves per CU: profiles/r04_memory_ceilings.json).  `roofline.peak` stays the 8 TB/s of the
# spec sheet; `stream_ceiling_GBps` is the number a memory-bound kernel can actually be held against.
STREAM_CEILINGS = os.path.join(ROOT, "profiles", "r04_memory_ceilings.json")
STREAM_MIX = {"nmf_cf_bwd_": "2:1", "nmf_cf_fwd_": "2:1", "mlp_chain_bwd_wgrad_": "3:1", "mlp_chain_fwd_": "1:1", "dgrad_": "3:1",
              "outproj_mlp_chain_fwd_": "1:1"}   # (2 tensors read, 4 written: the closest measured mix)


def stream_ceiling(timer_name):
    """(GB/s, mix) of the bare access pattern closest to this launch's read : write mix, or (None, None)"""
    try:
        d = json.load(open(STREAM_CEILINGS))
        mix = next((m for k, m in STREAM_MIX.items() if timer_name.startswith(k)), "2:1")
        best = max(r["GBps"] for r in d["streams"] if r["mix"] == mix and r["footprint_MiB_per_stream"] == 1024)
        return float(best), mix
    except Exception:
        return None, None
# timer key of a BASELINE-size (stage-0) launch -> kernel-name prefix in the PMC summary; the summary averages
# the launches with the largest grid of each kernel, i.e. the same stage-0 launches the key times
# (a tuple: the timer's launches are one launch of each named kernel — round 5: window 0 of the fused core's backward runs the
#  row-space kernel, window 1 the general one; the figure is then the mean over the kernels, like the timer's average launch)
PMC_KERNEL = {"nmf_cf_bwd_32x128x128x128": ("fz::nmf_cf_bwd_gram_kernel<", "fz::nmf_cf_bwd_tile_kernel<"),
              "nmf_cf_fwd_32x128x128x128": "fz::nmf_cf_fwd_tile_kernel<",
              "mlp_chain_bwd_32": "fz::gemm_chain_kernel<true",
              "mlp_chain_bwd_wgrad_32": "fz::gemm_chain_bwd_wg_kernel<",
              "dgrad_lnbwd_wgrad_32": "fz::gemm_dw_kernel<true",
              "dgrad_wgrad_32": "fz::gemm_dw_kernel<false",
              "mlp_chain_fwd_32": "fz::gemm_chain_kernel<false, 2, 2, float, true, false>",
              "outproj_mlp_chain_fwd_32": "fz::gemm_chain_kernel<false, 2, 2, float, true, true>"}


def pmc_traffic(timer_name, path=None):
    """(bytes per launch or None, traffic_source string).  Loud on stderr when the summary has no entry for
    the kernel the roofline line is about."""
    import hashlib
    PMC_TRAFFIC = path or globals()["PMC_TRAFFIC"]
    try:
        raw = open(PMC_TRAFFIC, "rb").read()
        d = json.loads(raw)
    except Exception as e:
        print(f"[bench] PMC traffic summary {PMC_TRAFFIC} unreadable: {e}", file=sys.stderr, flush=True)
        return None, f"missing: {os.path.relpath(PMC_TRAFFIC, ROOT)}"
    src = (f"{os.path.relpath(PMC_TRAFFIC, ROOT)} sha256:{hashlib.sha256(raw).hexdigest()[:12]} "
           f"measured_at_commit:{d.get('_meta', {}).get('commit', 'unknown')}")
    pre = PMC_KERNEL.get(timer_name)
    pres = pre if isinstance(pre, tuple) else ((pre,) if pre else ())
    per = [[v["traffic_bytes"] for k, v in d.items() if k.startswith(q) and isinstance(v, dict) and "traffic_bytes" in v] for q in pres]
    hits = [max(h) for h in per if h]
    if not hits or len(hits) != len(pres):
        print(f"[bench] NO PMC traffic entry for kernel {timer_name!r} (prefix {pre!r}) in {PMC_TRAFFIC}: "
              "roofline.traffic is null — re-run tools/pmc_traffic.py", file=sys.stderr, flush=True)
        return None, src + " (no entry for this kernel)"
    return int(sum(hits) / len(hits)), src


MODEL_KW = dict(in_channels=4, out_channels=3, spatial_size=(128, 128, 128),
                encoder_depth=(1, 1, 1, 1, 1), encoder_width=(32, 64, 128, 256, 512),
                strides=(1, 2, 2, 2, 2), decoder_depth=(1, 1, 1, 1), norm=ft.LayerNorm,
                reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                dropout=0.1)


def cpu_baseline_sample(batch=2, timed=3, all_threads=False):
    """CPU oracle (a port of the reference's CPU path, pinned to goldens generated by the imported reference) on THIS workload
    at THIS batch: whole training steps of the README Swin Factorizer — forward, DiceCE loss, backward, AdamW — on `batch` 128^3
    volumes (the benchmark's per-GPU batch), measured, not extrapolated: one warm-up step + `timed` timed steps on 32 threads
    (where ATen's CPU kernels run fastest at these sizes).  SURVEY.md §8d asks for torch.set_num_threads(os.cpu_count()): measured
    once in round 6 on the 256-thread host of the pool — 379.5 s per step against 34.4 s on 32 threads, 11 x SLOWER
    (profiles/r06_bench_default_with_all_threads_cpu_step.json) — so the default run keeps 32 threads and `--cpu-all-threads` adds
    that step (two more steps, ~13 minutes) for whoever wants it in the line.  BASELINE configs[0]-[2] ride along: the cfg-1 NMF
    forward, one FactorizerBlock forward+backward, one eval forward of the model (one timed run each, after the steps above).
    The long form is tools/cpu_baseline_full.py → profiles/rNN_cpu_baseline.json."""
    from oracle import cpu_ref as O
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(min(ncpu, 32))
    torch.manual_seed(0)
    model = ft.Factorizer(**{**MODEL_KW, "dropout": 0.0})
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    prm = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("u0", "v0"))}
    full = dict(sd)
    full.update(prm)
    cfg = dict(widths=MODEL_KW["encoder_width"], strides=MODEL_KW["strides"], reshape=dict(head_dim=8, patch_size=8),
               num_iters=5, solver="hals")
    opt = torch.optim.AdamW(list(prm.values()), lr=1e-4, weight_decay=1e-5)
    x = torch.rand(batch, 4, 128, 128, 128)
    tgt = (torch.rand(batch, 3, 128, 128, 128) > 0.5).float()
    # BASELINE cfg 1 (plumbing): ft.NMF((8,512), rank 2, 5 iterations, MU) forward on one matrix
    x1, u0, v0 = torch.rand(1, 8, 512), torch.rand(8, 2), torch.rand(512, 2)
    O.nmf_forward(x1, u0, v0, 5, "mu")
    t1 = time.perf_counter()
    for _ in range(20):
        O.nmf_forward(x1, u0, v0, 5, "mu")
    t_cfg1 = (time.perf_counter() - t1) / 20

    def one_step():
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = O.dice_ce_loss(O.factorizer_forward(x, full, cfg), tgt)
        loss.backward()
        opt.step()
        assert torch.isfinite(loss).item()
        return time.perf_counter() - t0

    warm = one_step()                       # first touch of every buffer, thread-pool start-up
    samples = [one_step() for _ in range(timed)]
    t = sum(samples) / len(samples)
    threads = torch.get_num_threads()
    # BASELINE configs[2] (eval forward of the same model) and configs[1] (one FactorizerBlock, forward + backward), same threads
    with torch.no_grad():
        t0 = time.perf_counter()
        O.factorizer_forward(x, full, cfg)
        t_cfg3 = time.perf_counter() - t0
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=32, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU, factorize=ft.NMF, rank=1,
                             num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
    bsd = {k: v.clone() for k, v in blk.state_dict().items()}
    bprm = {k: v.requires_grad_(True) for k, v in bsd.items() if not k.endswith(("u0", "v0"))}
    bfull = dict(bsd)
    bfull.update(bprm)
    bcfg = dict(reshape=dict(head_dim=8, patch_size=8), num_iters=5, solver="hals")
    xb = torch.rand(batch, 32, 128, 128, 128, requires_grad=True)
    gb = torch.rand(batch, 32, 128, 128, 128)

    def block_fb():
        t0 = time.perf_counter()
        torch.autograd.grad(O.factorizer_block(xb, bfull, "", bcfg), [xb] + list(bprm.values()), gb)
        return time.perf_counter() - t0
    t_cfg2 = block_fb()
    del xb, gb
    # the same training step with every host core
    t_all = None
    if all_threads and ncpu > threads:
        torch.set_num_threads(ncpu)
        one_step()
        t_all = one_step()
        torch.set_num_threads(threads)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except Exception:
        pass
    return {"value": batch / t, "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": "oracle (port of the reference's CPU path): whole training steps (forward + DiceCE + backward + AdamW) of "
                      f"the README Swin Factorizer at the benchmark's per-GPU batch ({batch} x 128^3): one warm-up step "
                      f"({warm:.1f} s) + {timed} timed steps ({', '.join(f'{v:.1f}' for v in samples)} s), mean {t:.1f} s, on "
                      f"{threads} threads of {ncpu} host CPUs ({cpu_model})"
                      + (f"; the same step with torch.set_num_threads({ncpu}): {t_all:.1f} s" if t_all is not None else
                         f"; with torch.set_num_threads({ncpu}) the step takes 11 x longer on this host type (379.5 s, measured once: "
                         "profiles/r06_bench_default_with_all_threads_cpu_step.json; --cpu-all-threads repeats it)"),
            "batch": batch, "seconds_sample": t, "seconds_warmup": warm, "seconds_timed": samples, "cpu_model": cpu_model,
            "seconds_step_all_host_threads": t_all, "host_cpus": ncpu,
            "cfg1_nmf_8x512_mu_r2_t5_fwd_us": round(t_cfg1 * 1e6, 1),
            "cfg2_block_fwd_bwd_seconds": round(t_cfg2, 2), "cfg3_model_eval_forward_seconds": round(t_cfg3, 2)}


def by_stage(table, nsteps, B, stage0_cols, dtype="f32"):
    """Where the step's kernel time is: per stage of the U-shape (a launch belongs to the stage of the finest tensor it
    touches: `cols` = batch x voxels, stage s has stage0_cols / 8^s) the summed launch time of the instrumented warm-up
    steps, the algorithmic bytes and matrix-core flops of those launches, and the minimum time either roof allows for that
    work: `hbm_min_ms` at 8 TB/s; `mfma_min_ms` with every launch family priced on the pipe it runs on — 157.3 TFLOP/s for
    the fp32 MFMA, dense bf16 peak / 6 = 419 TFLOP/s fp32-equivalent for the layers whose fp32 products are six bf16
    products (`split_products`).  `bound` = the larger of the two, `frac_of_binding_roof` = that minimum / the measured
    time.  `mfma_frac_of_fp32_peak` (flops / time / 157.3) is kept for comparison with round 2 and can pass 1 for the
    split-bf16 layers."""
    out = {}
    fam = {}
    for name, a in table.items():
        cols = a.get("cols", 0)
        s = 0
        while cols and cols * 8 ** s < stage0_cols and s < 8:
            s += 1
        key = "unattributed" if not cols else ("stage0" if s == 0 else "stage1" if s == 1 else "stage2-4")
        for k2, dst in ((key, out), ("gemm_family" if a.get("flops", 0) and not name.startswith(("wgrad", "mlp_chain", "outproj_mlp_chain", "dgrad_", "conv_k3", "upcat_")) else None, fam)):
            if k2 is None:
                continue
            d = dst.setdefault(k2, {"kernel_ms": 0.0, "GB": 0.0, "GFLOP": 0.0, "GFLOP_split_bf16": 0.0, "mfma_min_ms": 0.0, "launches": 0})
            d["kernel_ms"] += a["ms"] / nsteps
            d["GB"] += a["bytes"] / nsteps / 1e9
            d["GFLOP"] += a.get("flops", 0) / nsteps / 1e9
            if split_products(name, dtype):
                d["GFLOP_split_bf16"] += a.get("flops", 0) / nsteps / 1e9
            d["mfma_min_ms"] += a.get("flops", 0) / nsteps / 1e9 / mfma_roof_tflops(name, dtype)
            d["launches"] += a["calls"] // max(nsteps, 1)
    for dst in (out, fam):
        for d in dst.values():
            ms = max(d["kernel_ms"], 1e-9)
            d["hbm_min_ms"] = round(d["GB"] / HBM_PEAK_GBS * 1e3, 3)
            d["hbm_frac"] = round(d["GB"] / ms / HBM_PEAK_GBS * 1e3, 4)
            d["mfma_frac_of_fp32_peak"] = round(d["GFLOP"] / ms / FP32_MFMA_PEAK_TFLOPS, 4)
            d["mfma_frac"] = round(d["mfma_min_ms"] / ms, 4)
            d["bound"] = "mfma" if d["mfma_min_ms"] > d["hbm_min_ms"] else "hbm"
            d["frac_of_binding_roof"] = max(d["hbm_frac"], d["mfma_frac"])
            for k in ("kernel_ms", "GB", "GFLOP", "GFLOP_split_bf16", "mfma_min_ms"):
                d[k] = round(d[k], 3)
    out.update({"fz_gemm (1x1 layers, k2s2 convolutions: gemm_stream / gemm_bx / gemm_bxk / gemm_resident)": v for v in fam.values()})
    return out


NMF_FLOP_FWD = 95312  # HALS R=1 T=5 on one 8x512 matrix, forward (T·F_iter + F_recon)


def nmf_gflops(table, nsteps, B):
    """BASELINE.json's second metric, "NMF-iter GFLOP/s": algorithmic NMF flops of the fused
    matricize→NMF→inverse kernels ÷ their measured time (HIP events, per-kernel table).  Forward 95 312 flop per
    8x512 matrix; backward 3x (in-kernel recompute + reverse sweep).  Matrices per window launch at
    stage s of the README model: B·h·G."""
    stages = {"32x128x128x128": 4 * 4096, "64x64x64x64": 8 * 512, "128x32x32x32": 16 * 64,
              "256x16x16x16": 32 * 8, "512x8x8x8": 64}
    out = {}
    for kind, mult in (("fwd", 1.0), ("bwd", 3.0)):
        flops = ms = 0.0
        for shape, per_b in stages.items():
            a = table.get(f"nmf_cf_{kind}_{shape}")
            if not a:
                continue
            f = a["calls"] * B * per_b * NMF_FLOP_FWD * mult
            if shape == "32x128x128x128":
                out[f"{kind}_stage0"] = round(f / (a["ms"] * 1e-3) / 1e9, 1)
            flops += f
            ms += a["ms"]
        if ms > 0:
            out[f"{kind}_all_stages"] = round(flops / (ms * 1e-3) / 1e9, 1)
    out["unit"] = "GFLOP/s"
    out["flop_per_matrix_fwd"] = NMF_FLOP_FWD
    out["note"] = ("fused matricize+NMF+inverse launches (HBM-bound: 2.9 flop/B, SURVEY §8d); time includes the "
                   "gather/scatter of the same launch")
    return out



def other_configs(dev):
    """The other BASELINE.json configurations that fit one GPU, timed AFTER the headline's timed region so that the driver's
    one JSON line carries them: configs[1] FactorizerBlock fwd+bwd, configs[2] README model eval
    forward, configs[4] one GPU's share of the BraTS-shape stress case under bf16 autocast at its per-GPU batch of 4.
    Synthetic inputs, HIP events on the current stream; a few seconds in total."""
    import contextlib
    from torch import nn
    out = []

    def gpu_ms(fn, iters, warm):
        for _ in range(warm):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    def record(name, dtype, B, ms, **kw):
        out.append({"config": name, "dtype": dtype, "batch": B, "ms": round(ms, 3), "volumes_per_s": round(B / ms * 1e3, 2),
                    "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1), **kw})

    # configs[1]: single FactorizerBlock (C = 32, 128^3, head_dim 8, patch 8, HALS rank 1, 5 iterations), fwd + bwd, B = 2
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=32, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU, factorize=ft.NMF, rank=1,
                             num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0).to(dev)
    x = torch.rand(2, 32, 128, 128, 128, device=dev, requires_grad=True)
    g = torch.rand(2, 32, 128, 128, 128, device=dev)
    ps = [x] + list(blk.parameters())
    record("configs[1] FactorizerBlock C=32 128^3 fwd+bwd", "f32", 2, gpu_ms(lambda: torch.autograd.grad(blk(x), ps, g), 20, 5))
    del blk, x, g, ps
    # configs[2]: README Swin Factorizer, eval forward, B = 2
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    torch.manual_seed(0)
    model = ft.Factorizer(**MODEL_KW).to(dev).eval()
    x = torch.rand(2, 4, 128, 128, 128, device=dev)
    with torch.no_grad():
        record("configs[2] README Swin Factorizer eval forward", "f32", 2, gpu_ms(lambda: model(x), 15, 3))
    del model, x
    # configs[4], one GPU's share: 160x192x160, HALS rank 2, 10 iterations, patch (5,6,5) (p = 8 does not divide the shape,
    # SURVEY headline 5), bf16 autocast, per-GPU batch 4, training forward + loss + backward
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    torch.manual_seed(0)
    kw = dict(MODEL_KW, spatial_size=(160, 192, 160), reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}),
              rank=2, num_iters=10)
    model = ft.Factorizer(**kw).to(dev).train()
    x = torch.rand(4, 4, 160, 192, 160, device=dev)
    t = (torch.rand(4, 3, 160, 192, 160, device=dev) > 0.5).float()

    def fb():
        for p in model.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = ft.dice_ce_loss(model(x), t)
        loss.backward()
        return loss
    loss = fb()
    ms5 = gpu_ms(fb, 5, 2)
    # one more step with every native launch bracketed by HIP events: the per-kernel table and the roofline of the dominant kernel.
    # These launches are VALU-bound, not HBM-bound (8 x 150, HALS rank 2, 10 iterations = 119 760 flop per 4.8 KB
    # matrix forward, 3x that backward with the in-kernel recompute): priced against the fp32 VECTOR peak — 256 CUs x 4 SIMDs x
    # 16 lanes x 2 (packed fp32) x 2 flop x 2.4 GHz = 157.3 TFLOP/s, the same number as the fp32 matrix peak.
    tm = Fn.KernelTimer()
    Fn.set_timer(tm)
    fb()
    Fn.set_timer(None)
    agg = tm.summary()
    tot = sum(a["ms"] for a in agg.values())
    M5, N5, R5, T5 = 8, 150, 2, 10
    f_iter = 4 * M5 * N5 * R5 + 2 * (M5 + N5) * R5 * R5 + 2 * (M5 + N5) * R5 * (R5 - 1)
    f_fwd = T5 * f_iter + 2 * M5 * N5 * R5                       # 119 760
    roof5 = None
    nmf_rows = {k: a for k, a in agg.items() if k.startswith(("nmf_pcf_fwd", "nmf_pcf_bwd"))}
    if nmf_rows:
        name, a = max(nmf_rows.items(), key=lambda kv: kv[1]["ms"])
        C5 = int(name.split("_")[3].split("x")[0])
        nmat = a["cols"] * C5 // 8 // N5                          # matrices per launch: batch x heads x patches
        flop = nmat * f_fwd * (3 if "_bwd_" in name else 1)
        avg_ms = a["ms"] / a["calls"]
        tfl = flop / (avg_ms * 1e-3) / 1e12
        roof5 = {"bound": "valu", "kernel": name, "achieved": round(tfl, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(tfl / FP32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(avg_ms, 3), "launches": a["calls"],
                 "matrices_per_launch": nmat, "flop_per_matrix": f_fwd * (3 if "_bwd_" in name else 1),
                 "flop_formula": "SURVEY.md 8(d): F = T*F_iter(HALS) + F_recon = 119 760 at 8x150, R 2, T 10; backward 3F",
                 "hbm_GBps": round(a["bytes"] / a["calls"] / (avg_ms * 1e-3) / 1e9, 1),
                 "share_of_step": round(a["ms"] / ms5, 4),
                 "nmf_launches_ms": {k: [v["calls"], round(v["ms"], 3)] for k, v in sorted(nmf_rows.items(), key=lambda kv: -kv[1]["ms"])},
                 "instruction_mix": "profiles/r06_cfg5.md"}
    table5 = {k: [v["calls"], round(v["ms"], 3)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:14]}
    record("configs[4] BraTS-shape stress 160x192x160, HALS R2 T10, patch (5,6,5), fwd+loss+bwd", "bf16 activations (autocast), fp32 "
           "parameters / statistics / NMF internals", 4, ms5, loss_finite=bool(torch.isfinite(loss).item()), roofline=roof5,
           native_kernels_ms_total=round(tot, 2), top_kernels_calls_ms=table5)
    del model, x, t
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-threads", action="store_true",
                    help="cpu_baseline: also time the training step with torch.set_num_threads(os.cpu_count()) (minutes on a 256-thread host)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BASELINE configs[1], [2], [4] timings appended to the JSON line as `other_configs`")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 (the reference's precision: the headline line) or bf16 = torch.autocast(bfloat16): bf16 "
                         "activation storage, fp32 parameters / statistics / accumulation / NMF internals (BASELINE configs[4] mode)")
    ap.add_argument("--hp-stream", action="store_true",
                    help="run the step on a high-priority HIP stream (the side stream of the deferred weight gradients "
                         "keeps the default, lower priority)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the hook-launched bucket all-reduces and "
                         "finish() even with one rank (the N = 1 line then executes the N > 1 code path)")
    ap.add_argument("--no-late-join", action="store_true",
                    help="await the side-stream weight gradients at the end of every block backward "
                         "(factorizer_amd/pointwise.py:_LateJoin)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks "
                         f"(WORLD_SIZE={world})")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU")
    dev_index = local_rank % ndev  # one process per GPU; the modulo only matters for the 1-GPU gloo self-test
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    json_fd = None
    if use_dist:
        # RCCL prints its version banner to the C-level stdout: keep the ONE JSON line alone there by
        # pointing fd 1 at stderr for the life of the process group and writing the line to the saved fd
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("FZ_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    torch.manual_seed(0)
    model = ft.Factorizer(**MODEL_KW).to(dev).train()
    sync = FlatGradSync(model, num_buckets=4, overlap=True, late_wgrad_join=not args.no_late_join,
                        force_collectives=args.force_dist)
    sync.broadcast_state(0)
    # AdamW of the recipe (train.yaml:72-76: lr 1e-4, wd 1e-5) as ONE kernel over the flat parameter /
    # gradient / moment buffers (csrc/optim.hip); the gradient buffer is the one RCCL reduces in place
    opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5, flat_grad=sync.flat, grad_views=sync.views)
    B = args.batch_per_gpu
    torch.manual_seed(1234 + rank)
    x = torch.rand(B, 4, 128, 128, 128, device=dev)
    target = (torch.rand(B, 3, 128, 128, 128, device=dev) > 0.5).float()

    if args.hp_stream:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    import contextlib
    amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if args.dtype == "bf16" else contextlib.nullcontext

    def step():
        sync.zero_grad()
        with amp():
            loss = ft.dice_ce_loss(model(x), target)   # the recipe's DiceCELoss(sigmoid, squared_pred), train.yaml:67-70
        loss.backward()
        scale = sync.finish(average=False)          # SUM stays in the buffer; 1/world is applied by the optimizer kernel
        opt.step(grad_scale=scale)
        return loss

    # Warm-up steps run with EVERY native launch bracketed by HIP events: that pass yields the per-kernel
    # table and names the dominant kernel.  In the timed region only that kernel is bracketed (two
    # event records per launch cost ~1 us of stream time each; ~940 of them are ~4 % of a step).
    wtimer = Fn.KernelTimer()
    wsteps = 0
    wstep_events = []   # (start, end) of each fully instrumented step on the launch stream: its device time incl. everything the table does not see
    for i in range(args.warmup):
        # the very first step pays one-time costs (code-object loading, first-touch allocations): it is
        # left out of the per-kernel table when there is a second warm-up step to take its place
        if i == 0 and args.warmup > 1:
            step()
            continue
        Fn.set_timer(wtimer)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        step()
        ev1.record()
        wstep_events.append((ev0, ev1))
        wsteps += 1
    Fn.set_timer(None)
    wagg = wtimer.summary() if wsteps > 0 else {}
    dominant = max(wagg.items(), key=lambda kv: kv[1]["ms"])[0] if wagg else None
    timer = Fn.KernelTimer(only=None if dominant is None else {dominant})
    Fn.set_timer(timer)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    Fn.set_timer(None)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss).item(), "loss is not finite"

    copy_gbs = None
    if rank == 0:
        # context for the roofline fraction: what a plain device-to-device copy of one stage-0 activation
        # (537 MB read + 537 MB written) reaches on THIS box (4.6-5.3 TB/s; the 8 TB/s spec is not reachable)
        src = torch.empty(2 * 32 * 128 ** 3, device=dev)
        dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = round(5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst
    if rank == 0:
        agg = timer.summary()
        roof = None
        if agg:
            name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
            avg_ms = a["ms"] / a["calls"]
            gbs = a["bytes"] / a["calls"] / (avg_ms * 1e-3) / 1e9
            # (the counter passes are committed per command: the fp32 headline and the --dtype bf16 secondary line)
            traffic, traffic_source = pmc_traffic(name, PMC_TRAFFIC if args.dtype == "f32" else PMC_TRAFFIC_BF16)
            ceil_gbs, ceil_mix = stream_ceiling(name)
            # which roof binds: the one with the larger minimum time for this launch's algorithmic work
            fpb = next((v for k, v in MFMA_FLOP_PER_BYTE.items() if name.startswith(k)), 0.0)
            tflops = fpb * gbs / 1e3
            mpeak = round(mfma_roof_tflops(name, args.dtype), 1)   # the pipe this kernel's products run on
            if fpb * HBM_PEAK_GBS / 1e3 > mpeak:     # arithmetic intensity above that pipe's ridge
                head = {"bound": "mfma", "kernel": name, "achieved": round(tflops, 2), "peak": mpeak,
                        "unit": "TFLOP/s", "frac": round(tflops / mpeak, 4),
                        "flop_per_algorithmic_byte": fpb, "hbm_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
            else:
                head = {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
            if head["bound"] == "hbm" and ceil_gbs:
                es = 1.0 if args.dtype == "f32" else 1.0   # (ceilings are in bytes: storage type does not matter)
                head["stream_ceiling_GBps"] = ceil_gbs
                head["stream_ceiling_mix"] = ceil_mix
                head["frac_of_stream_ceiling"] = round(gbs * es / ceil_gbs, 4)
            roof = {**head, "traffic": traffic,
                    "traffic_source": traffic_source,
                    "avg_launch_ms": round(avg_ms, 4), "launches": a["calls"],
                    "algorithmic_bytes_per_launch": a["bytes"] // a["calls"],
                    "share_of_step": round(a["ms"] / (elapsed * 1e3), 4),
                    "device_copy_GBps_this_box": copy_gbs,
                    "timed_with_events": "dominant kernel only (chosen from the fully instrumented warm-up steps)"
                    if dominant is not None else "all native launches",
                    "native_kernels_ms_per_step": {k: round(v["ms"] / (wsteps if wagg else max(args.steps, 1)), 3) for k, v in (wagg or agg).items()},
                    "native_kernels_GBps": {k: round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 0) for k, v in (wagg or agg).items()},
                    "native_kernels_table_from": "warm-up steps" if wagg else "timed steps",
                    "by_stage": by_stage(wagg or agg, wsteps if wagg else max(args.steps, 1), B, B * 128 ** 3, args.dtype)}
            # what the per-kernel table does NOT see: framework kernels (loss glue, gradient packing), copies, idle gaps of the
            # stream — the timed step minus the summed device time of the library's launches in the instrumented steps
            lib_ms = sum(roof["native_kernels_ms_per_step"].values())
            roof["by_stage"]["library_kernels_ms_per_step"] = round(lib_ms, 3)
            if wstep_events:
                inst_ms = sum(a.elapsed_time(b) for a, b in wstep_events) / len(wstep_events)
                roof["by_stage"]["instrumented_step_ms"] = round(inst_ms, 3)
                roof["by_stage"]["other_device_time_ms"] = round(inst_ms - lib_ms, 3)
                roof["by_stage"]["other_device_time_note"] = (
                    "device time of the fully instrumented warm-up steps (every launch bracketed by events: ~0.7 ms slower than a timed step) "
                    "minus the library's summed launch time: framework kernels (8 launches, 0.05 ms per step — dropout and position "
                    "embedding of the bottleneck, loss glue, gradient packing; NO device-to-device copy runs inside a step: "
                    "profiles/r06_step_copies.md) + the event records themselves + gaps between launches")
        out = {
            "metric": "volumes/sec fwd+bwd, Swin Factorizer 128^3",
            "value": round(world * B * args.steps / elapsed, 4),
            "unit": "volumes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16",
            "dtype_note": ("fp32 storage, fp32 accumulation, fp32-accurate products throughout (the reference runs amp: false, "
                           "train.yaml:34): layers with a reduction length >= 64 and all weight gradients form each fp32 product "
                           "from six exact bf16 MFMA products of three-level operand splits — error <= the fp32 MFMA's own, "
                           "profiles/r03_bx6_accuracy.json, tests/test_gpu_bx.py; FZ_GEMM_BX=0 puts every product back on "
                           "v_mfma_f32_32x32x2_f32" if args.dtype == "f32" else
                           "torch.autocast(bfloat16): bf16 activation storage; parameters, LayerNorm statistics, MFMA "
                           "accumulation, weight gradients and the NMF iteration (U, V, Gram, eps) fp32"),
            "data": "synthetic",
            "config": {"workload": "Swin Factorizer (in4,out3,128^3,widths 32-512,d8,p8,HALS R1 T5) "
                                   "training step fwd+bwd+AdamW (BASELINE configs[3])",
                       "global_batch": world * B, "batch_per_gpu": B, "parallelism": f"dp{world}"},
            "roofline": roof,
            "nmf_iter_gflops": nmf_gflops(wagg or agg, wsteps if wagg else max(args.steps, 1), B),
            "rccl": ("process group 'nccl' (RCCL), hook-launched bucket all-reduces executed"
                     if use_dist and os.environ.get("FZ_BENCH_BACKEND", "nccl") == "nccl" else
                     ("gloo" if use_dist else "not initialised (single rank; --force-dist runs it)")),
        }
        if world == 1 and not args.no_other_configs and args.dtype == "f32":
            del model, opt, sync, x, target
            out["other_configs"] = other_configs(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sample(batch=B, all_threads=args.cpu_all_threads)
        if json_fd is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
