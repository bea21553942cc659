#!/usr/bin/env python3
"""bench.py -- edit+eval cycles/sec, BLIP-2-OPT-2.7B + FT_VL on synthetic EVQA-shaped inputs
(BASELINE.json config[1]: 1000 synthetic edits, bf16, 1xMI355X; N ranks = weak scaling, one
process per GPU, splits sharded with no data-path collective and one gather of score rows).

A "step" = one batch of --cycles-per-step independent edit+eval cycles through the batched HIP
engine (devqa_amd.batched.BatchedEditEval): 9 pre-edit locality probes -> FT_VL edit (<= 25 fused
AdamW steps, early stop enabled) -> 12 post-edit probes, per cycle.  Inputs (pre-processed pixel
values, token ids) are resident in HBM/host lists before the timed region starts.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# deduplicated algorithmic work per cycle (A_min) and its GEMM-shaped share
A_MIN_TFLOP_PER_CYCLE = 3.81
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def build_full_model(dev, seed, layers=None, threads=16, keep_host_copy=False):
    """BLIP-2-OPT-2.7B dims, numpy-seeded synthetic weights ('opt' recipe), generated in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    import devqa_amd  # noqa: F401
    from devqa_amd import blip2_spec
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.synth import param_init
    cfg = blip2_spec.BLIP2_OPT_2_7B if layers is None else blip2_spec.scaled_spec(*layers)
    model = Blip2Native(cfg, dev, "bf16")
    names = list(model._shapes.keys())
    kept = {} if keep_host_copy else None
    with ThreadPoolExecutor(threads) as ex:
        futs = {n: ex.submit(param_init, n, model._shapes[n], seed, "opt") for n in names}
        for n in names:
            arr = futs.pop(n).result()
            model.load_named_tensors(lambda _n, a=arr: torch.from_numpy(a), names=[n], refresh=False)
            if kept is not None:
                kept[n] = arr
    model.refresh_derived(force=True)
    return model, cfg, kept


def cpu_baseline(cfg, seed, threads, arrays=None):
    """The oracle (CPU restatement of the reference path, fp32, full BLIP-2-OPT-2.7B dims) timed on a
    bounded sample: ONE image encode (ViT-g + Q-Former), ONE decoder forward at T=48 and ONE FT_VL step
    (forward + backward onto layers.31.fc2.weight + torch.optim.AdamW).  A reference-style cycle executes
    40 encodes + 46 decoder forwards (25 of them inside FT steps) -- SURVEY.md 3.1 -- so
    cycles/s = 1 / (40*t_enc + 21*t_dec + 25*t_step)."""
    from oracle import devqa_oracle as O
    from devqa_amd import blip2_spec
    from devqa_amd.synth import param_init
    torch.set_num_threads(threads)
    shapes = blip2_spec.param_shapes(cfg)
    t0 = time.time()
    if arrays is None:
        arrays = {n: param_init(n, s, seed, "opt") for n, s in shapes.items()}
    w = {n: torch.from_numpy(arrays[n]) for n in shapes}
    m = O.OracleBlip2(w, cfg, None, copy=False)
    gen_s = time.time() - t0
    g = torch.Generator().manual_seed(0)
    pix = torch.randn(1, 3, 224, 224, generator=g)
    with torch.no_grad():
        t0 = time.time()
        it = m.image_tokens(pix)
        t_enc = time.time() - t0
        emb = torch.cat([it, torch.randn(1, 16, it.shape[-1], generator=g) * 0.05], 1)
        msk = torch.ones(1, 48, dtype=torch.long)
        t0 = time.time()
        m.get_llm_outpt({"inputs_embeds": emb, "attention_mask": msk})
        t_dec = time.time() - t0
    name = "language_model.model.decoder.layers.%d.fc2.weight" % (cfg["text_config"]["num_hidden_layers"] - 1)
    p = m.w[name].clone().requires_grad_(True)
    m.w[name] = p
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=0)
    y = torch.randint(4, 50272, (1, 3), generator=g)
    mk = torch.tensor([[1, 1, 0]])
    t0 = time.time()
    loss = O.label_loss(m.get_llm_outpt({"inputs_embeds": emb, "attention_mask": msk}), y, mk)
    loss.backward()
    opt.step()
    t_step = time.time() - t0
    cyc = 40 * t_enc + 21 * t_dec + 25 * t_step
    return {"value": 1.0 / cyc, "unit": "cycles/s", "cores": threads, "kind": "port",
            "sample": "1 ViT-g+Q-Former encode (%.2fs) + 1 OPT-2.7B forward T=48 (%.2fs) + 1 FT step fwd+bwd+AdamW (%.2fs), "
                      "fp32 torch CPU oracle at full BLIP-2-OPT-2.7B dims; cycle = 40 enc + 21 fwd + 25 steps = %.1fs"
                      % (t_enc, t_dec, t_step, cyc), "weight_gen_s": round(gen_s, 1)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py; FETCH_SIZE and
    WRITE_SIZE are collected in separate profiler runs of this same command, so the figure is read from the newest
    profiles/*pmc_traffic.json rather than measured inside the timed run).  None when no pass covers the kernel."""
    import glob
    key = kernel.replace(" ", "")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            for r in json.load(open(f))["kernels"]:
                if r["kernel"].replace(" ", "").startswith(key):
                    return {"hbm_bytes_per_launch": r["hbm_bytes_per_launch"], "read": r["read_bytes_per_launch"],
                            "write": r["write_bytes_per_launch"], "source": os.path.basename(f)}
        except (OSError, KeyError, ValueError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cycles-per-step", type=int, default=60)
    ap.add_argument("--seed", type=int, default=20251121)
    ap.add_argument("--layers", type=str, default=None, help="debug only: 'v,q,t' layer counts (INVALID as a benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="run the two stages of every batch back to back on one stream")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import devqa_amd  # noqa: F401
    from devqa_amd import lib
    from devqa_amd.batched import BatchedEditEval
    from devqa_amd.dist import gather_score_rows, init_from_env
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.synth import IdTokenizer, evqa_cycles, synth_image_u8
    import torch.distributed as dist

    # DEVQA_FORCE_DIST=1: build the process group (RCCL) even for one rank, so that a 1-GPU box walks the N > 1 branches below
    rank, world = init_from_env(min_world=1 if os.environ.get("DEVQA_FORCE_DIST") else 2)
    use_dist = dist.is_initialized()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(dev)
    lib.load()
    layers = None if args.layers is None else tuple(int(x) for x in args.layers.split(","))
    t0 = time.time()
    want_cpu = (not args.no_cpu_baseline) and world == 1
    model, cfg, host_arrays = build_full_model(dev, args.seed, layers, keep_host_copy=want_cpu)
    vllm = BLIP2OPTForEdit(None, dev, model=model, tokenizer=IdTokenizer())
    ft_cfg = FTvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "ft_vl", "blip2-opt-2.7b.yaml"))
    ft_cfg.layers = [cfg["text_config"]["num_hidden_layers"] - 1]
    editor = FTvl(vllm, ft_cfg, dev)
    be = BatchedEditEval(editor, cycles_per_batch=args.cycles_per_step)
    build_s = time.time() - t0

    # ---- synthetic inputs, resident before the timed region (pixel values pre-processed into HBM) ----
    E, K, W = args.cycles_per_step, args.steps, args.warmup
    n_cyc = (K + W) * E
    proc = vllm.image_processor
    img_cache = {}

    def image_of(s, tag):
        key = (s, tag)
        if key not in img_cache:
            gs = rank * n_cyc + s  # distinct samples per rank (weak scaling: per-GPU work fixed)
            img_cache[key] = torch.from_numpy(proc(synth_image_u8(gs, tag, cfg["vision_config"]["image_size"], args.seed))).to(dev)
        return img_cache[key]
    cycles = evqa_cycles(n_cyc, cfg["text_config"]["vocab_size"], args.seed + 7919 * rank, image_of)
    from devqa_amd.batched import copy_sample
    batches = [([copy_sample(c) for c in cycles[i * E:(i + 1) * E]], [c for c in cycles[i * E:(i + 1) * E]])
               for i in range(K + W)]
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if W:
        be.run_batches(batches[:W], pipelined=not args.no_pipeline)
    for k in be.stats:
        be.stats[k] = 0
    barrier()
    lib.profile_gemm(1)
    t0 = time.time()
    outs, metas = [], []
    for o, mt in be.run_batches(batches[W:W + K], pipelined=not args.no_pipeline):
        outs += o
        metas += mt
    barrier()
    elapsed = time.time() - t0
    lib.profile_gemm(0)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rows = BatchedEditEval.score_rows(outs, metas, rank * K * E)
        gather_score_rows(rows, world * K * E, rank, world, torch.device(dev))  # the single RCCL gather
    prof = lib.profile_gemm_read()
    steps_mean = be.stats["steps"] / max(be.stats["cycles"], 1)

    if rank == 0:
        total_cycles = world * K * E
        value = total_cycles / elapsed
        names = ["gemm_bf16_tn_kernel<32,128,1,4>", "gemm_bf16_glds_kernel<64,128,2,2>", "gemm_bf16_glds_kernel<128,128,2,2>",
                 "gemm_bf16_pp_kernel"]
        dom = max(range(4), key=lambda i: prof[i][0])
        g_ms = sum(p[0] for p in prof)
        g_fl = sum(p[1] for p in prof)
        g_n = sum(p[2] for p in prof)
        covered_cycles = K * E * (g_n / max(1, g_n))  # all launches of the timed region unless the event pool filled
        dms, dfl, dn = prof[dom]
        # algorithmic FLOPs of the dominant kernel's launches: its share of the executed GEMM FLOPs scaled to A_min
        exec_per_cycle = g_fl / (K * E) / 1e12
        alg_scale = min(1.0, A_MIN_TFLOP_PER_CYCLE / exec_per_cycle) if exec_per_cycle > 0 else 0.0
        achieved = (dfl * alg_scale / 1e12) / (dms / 1e3) if dms > 0 else 0.0
        traffic = pmc_traffic(names[dom])
        out = {
            "metric": "edit+eval cycles/sec, BLIP-2 FT_VL EVQA", "value": round(value, 3), "unit": "cycles/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BLIP-2-OPT-2.7B + FT_VL, %d synthetic EVQA-shaped edit+eval cycles per GPU "
                                   "(%d per step), bf16 weights/activations, fp32 master + AdamW state for the edited "
                                   "fc2 matrix, early stop enabled" % (K * E, E),
                       "cycles_per_step": E, "cycles_total": total_cycles, "mean_ft_steps": round(steps_mean, 2),
                       "layers": "39/12/32" if layers is None else args.layers, "sharding": "splits block-partitioned, 1 gather"},
            "roofline": {"bound": "mfma", "kernel": names[dom], "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "traffic": None if traffic is None else traffic["hbm_bytes_per_launch"], "traffic_detail": traffic,
                         "avg_launch_us": round(1e3 * dms / max(dn, 1), 2), "launches": int(dn),
                         "executed_tflops": round((dfl / 1e12) / (dms / 1e3), 1) if dms > 0 else 0.0,
                         "all_gemm_executed_tflops": round((g_fl / 1e12) / (g_ms / 1e3), 1) if g_ms > 0 else 0.0,
                         "gemm_time_frac_of_step": round((g_ms / 1e3) / elapsed, 3),
                         "executed_tflop_per_cycle": round(exec_per_cycle, 3), "a_min_tflop_per_cycle": A_MIN_TFLOP_PER_CYCLE,
                         "path_frac_of_mfma_peak": round(value / world * A_MIN_TFLOP_PER_CYCLE / MFMA_PEAK_TFLOPS, 4)},
            "phase_s": {k: round(v, 3) for k, v in be.stats.items() if k.startswith("t_")},
            "build_s": round(build_s, 1),
        }
        if want_cpu:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 64)
            del model, vllm, editor, be
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(cfg, args.seed, threads, host_arrays)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
