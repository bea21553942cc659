#!/usr/bin/env python3
"""bench.py -- edit+eval cycles/sec, BLIP-2-OPT-2.7B + FT_VL on synthetic EVQA-shaped inputs (BASELINE.json config[1]).

    python bench.py --gpus N --steps K --warmup W

A "step" = one batch of --cycles-per-step independent edit+eval cycles through the batched HIP engine
(devqa_amd.batched.BatchedEditEval): 9 pre-edit locality probes -> FT_VL edit (<= 25 fused AdamW steps, early stop enabled) -> 12
post-edit probes, per cycle.  Inputs (pre-processed pixel values, token ids) are resident in HBM / host lists before the timed
region starts.  One process per GPU: with --gpus N > 1 and no WORLD_SIZE in the environment this script starts N child ranks
itself (fresh processes, started BEFORE anything touches a GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and
relays rank 0's JSON line; under torchrun it is one of the ranks.  Ranks shard the edit stream with no data-path collective and
meet in ONE RCCL all-gather of per-cycle score rows.

  --scaling weak    (default) every rank runs K steps of E cycles: per-GPU work fixed.
  --scaling strong  K * E cycles in total, block-partitioned over the ranks ("-sen 1000 edit stream shards across the 8 GPUs").
  --ffn sparse|dense|both   weight recipe of the decoder FFN.  "sparse" (the headline): devqa_amd.synth style "opt" -- fc1 bias
                    -0.3 and a narrow fc1 so that ~2 % of the ReLU units fire, as in a trained OPT; the FT loop then runs on the
                    active columns only.  "dense": SURVEY.md 8(d)'s recipe verbatim (N(0, 0.02) weights, biases 0): half of
                    the units fire, no compaction.  "both" (default): the sparse headline plus a short dense leg reported beside it.
  --selftest-cpu    no GPU, gloo: walks the launcher / sharding / collective / JSON code with fake cycles (tests/test_dist_cpu.py).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# deduplicated algorithmic work per cycle (A_min)
A_MIN_TFLOP_PER_CYCLE = 3.81
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
SPARSE_RECIPE = ("synthetic weights, numpy recipe 'opt' (devqa_amd/synth.py): decoder fc1 bias -0.3 and fc1 ~ N(0, 0.15/sqrt(d)) so that "
                 "~2% of the ReLU FFN units fire (trained-OPT-like sparsity); the FT loop carries the active columns only")
DENSE_RECIPE = "SURVEY 8(d) recipe verbatim: every weight N(0, 0.02), LayerNorm weight 1, all biases 0 (dense ReLU FFN, no column compaction)"


PROF = {"frac": 1.0, "steps": []}      # share of the last timed region that carried the slot profiler's event pairs (timed_leg)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cycles-per-step", type=int, default=127,
                    help="cycles per batch; 127 -> 508 images through the vision encoder in one call (130556 rows = 510 row tiles of 256)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="which leg is the headline `value`; the other leg is reported beside it (`strong` / `weak` object)")
    ap.add_argument("--strong-cycles", type=int, default=1000, help="length of the edit stream of the strong-scaling leg (north_star: -sen 1000)")
    ap.add_argument("--no-second-leg", action="store_true", help="skip the leg that is not the headline")
    ap.add_argument("--dump-rows", type=str, default=None, help="rank 0 writes the gathered [n, 16] score rows of the strong leg here (.npy)")
    ap.add_argument("--ffn", choices=["sparse", "dense", "both"], default="both")
    ap.add_argument("--dense-steps", type=int, default=0, help="steps of the dense leg under --ffn both (default max(3, steps // 3))")
    ap.add_argument("--seed", type=int, default=20251121)
    ap.add_argument("--layers", type=str, default=None, help="debug only: 'v,q,t' layer counts (INVALID as a benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cycles", type=int, default=1, help="fully timed reference-style oracle cycles of the cpu_baseline leg")
    ap.add_argument("--no-hbm-micro", action="store_true", help="skip the dense-AdamW / cosine top-k HBM measurements")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-depth parity block (GPU bf16 / fp32 mode vs the oracle cycle of cpu_baseline)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs #3-#5 legs (LLaVA + FT_VL, BLIP-2 + MEND_VL, MiniGPT-4 + IKE_VL)")
    ap.add_argument("--config-cycles", type=str, default="32,64,16", help="cycles of the three configs legs (llava_ft, blip2_mend, minigpt4_ike)")
    ap.add_argument("--profile-every", type=int, default=4, help="the slot profiler's event pairs bracket every N-th step of the timed region (1: every step)")
    ap.add_argument("--no-pipeline", action="store_true", help="run the two stages of every batch back to back on one stream")
    ap.add_argument("--host-pixels", action="store_true",
                    help="measurement only (never `value`): the pre-processed pixel values stay in pinned HOST memory and cross PCIe inside the timed "
                         "region -- the rate a caller sees who hands over host buffers (DESIGN 7)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--selftest-cpu", action="store_true")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torchrun environment
# ---------------------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start n fresh child processes of this script (one rank per GPU), wait for all of them, relay rank 0's stdout.  The parent
    never initialises a GPU and never replaces itself (no exec).  Exit code: 0 only if every rank exited 0."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while True:     # a rank that dies would leave the others blocked in a collective until the store times out: end them at once
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=30)
    sys.stdout.write("".join(c for c in chunks if c))
    sys.stdout.flush()
    if failed or any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        return 1
    return 0


def split_batches(n, E):
    """Sizes of the batches a rank cuts its block of n cycles into: at most E cycles each, and AT LEAST TWO batches whenever n > 1, so
    that the software pipeline (stage A of batch i + 1 queued under stage B of batch i, batched.py) has something to overlap even when
    a rank's whole block (1000 cycles / 8 ranks = 125) would fit one batch; sizes differ by at most one."""
    if n <= 0:
        return []
    nb = max(-(-n // E), 2 if n > 1 else 1)
    return [n // nb + (1 if i < n % nb else 0) for i in range(nb)]


def instrumented_iters(n_it, every, pipelined=True):
    """Iterations of the pipelined batch loop whose launches carry the slot profiler's event pairs: every `every`-th one, never the last (it has no
    stage A of a next batch, i.e. is not a whole step's launches).  Empty set = instrument everything (short runs, --profile-every 1, no pipeline)."""
    if every <= 1 or not pipelined or n_it < 2:
        return set()
    return set(i for i in range(n_it) if i % every == every // 2 and i + 1 < n_it)


def build_threads(world):
    """Host threads of the synthetic-weight generator: 8 ranks x 16 threads on a 64-core host would oversubscribe it 2x during the
    (untimed) model build; share the cores between the ranks of this node."""
    return max(2, min(16, (os.cpu_count() or 16) // max(world, 1)))


# ---------------------------------------------------------------------------------------------------------------------------------
# model / cpu baseline / HBM micro-measurements
# ---------------------------------------------------------------------------------------------------------------------------------
def build_full_model(dev, seed, layers=None, threads=16, keep_host_copy=False, style="opt"):
    """BLIP-2-OPT-2.7B dims, numpy-seeded synthetic weights, generated in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    import devqa_amd  # noqa: F401
    from devqa_amd import blip2_spec
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.synth import param_init
    cfg = blip2_spec.BLIP2_OPT_2_7B if layers is None else blip2_spec.scaled_spec(*layers)
    model = Blip2Native(cfg, dev, "bf16")
    names = list(model._shapes.keys())
    kept = {} if keep_host_copy else None
    with ThreadPoolExecutor(threads) as ex:
        futs = {n: ex.submit(param_init, n, model._shapes[n], seed, style) for n in names}
        for n in names:
            arr = futs.pop(n).result()
            model.load_named_tensors(lambda _n, a=arr: torch.from_numpy(a), names=[n], refresh=False)
            if kept is not None:
                kept[n] = arr
    model.refresh_derived(force=True)
    return model, cfg, kept


def cpu_baseline(cfg, seed, threads, arrays, cycles, n_cycles, style="opt", trace=False, cache_images=False):
    """The oracle (CPU restatement of the reference path, fp32, full BLIP-2-OPT-2.7B dims) timed on `n_cycles` COMPLETE
    reference-style cycles (oracle.devqa_oracle.faithful_cycle_pretokenized: 9 pre-edit forwards, <= 25 x [image encode + forward +
    backward + torch.optim.AdamW], 12 post-edit forwards; nothing cached -- SURVEY.md 3.1) after one un-timed warm-up call (one
    image encode + one decoder forward), on the same synthetic cycles the GPU ran."""
    import torch
    from oracle import devqa_oracle as O
    from devqa_amd import blip2_spec
    from devqa_amd.synth import param_init
    torch.set_num_threads(threads)
    shapes = blip2_spec.param_shapes(cfg)
    t0 = time.time()
    if arrays is None:
        arrays = {n: param_init(n, s, seed, style) for n, s in shapes.items()}
    w = {n: torch.from_numpy(arrays[n]) for n in shapes}
    m = O.OracleBlip2(w, cfg, None, copy=False)
    gen_s = time.time() - t0
    wname = "language_model.model.decoder.layers.%d.fc2.weight" % (cfg["text_config"]["num_hidden_layers"] - 1)

    def host(c):   # device pixel tensors -> host arrays (inputs only)
        def conv(it):
            return dict(it, image=None if it["image"] is None else it["image"].detach().float().cpu().numpy())
        return {"requests": [conv(c["requests"][0])], "generality": {k: [conv(v[0])] for k, v in c["generality"].items()},
                "locality": {k: [conv(v[0])] for k, v in c["locality"].items()}}
    cyc = [host(c) for c in cycles[:n_cycles]]
    with torch.no_grad():   # warm-up: thread pools, allocator, page-in of the weights
        x, _, _ = O._pretok_xym(m, cyc[0]["requests"][0]["prompt"], cyc[0]["requests"][0]["image"], cyc[0]["requests"][0]["target_new"])
        m.get_llm_outpt(x, None)
    t0 = time.time()
    info = [O.faithful_cycle_pretokenized(m, c, wname, 25, 1e-3, 0.0, trace=trace, cache_images=cache_images) for c in cyc]
    dt = time.time() - t0
    cpu_baseline.last_traces = info
    return {"value": len(cyc) / dt, "unit": "cycles/s", "cores": threads, "kind": "port",
            "sample": "%d complete reference-style cycle(s) (%d image encodes, %d decoder forwards, %d FT steps with backward + "
                      "torch.optim.AdamW each), fp32 torch CPU oracle at full BLIP-2-OPT-2.7B dims, %.1f s after one warm-up "
                      "encode + forward" % (len(cyc), info[0]["encodes"], info[0]["forwards"], info[0]["steps"], dt),
            "seconds_per_cycle": round(dt / len(cyc), 2), "weight_gen_s": round(gen_s, 1)}


def hbm_micro(dev, d_out=2560, d_in=10240):
    """HBM-regime kernels measured on their own with HIP events (lib.profile slots): the DENSE ft_adamw_step sweep in the form the path
    runs it (first moment kept as its rank-L factors: w and v cross HBM, 4 x 4 x 2560 x 10240 = 419 MB per edit-step; the form with a
    first-moment matrix, 6 x 4 = 629 MB, beside it) and cosine top-k over the VLKEB-shaped corpus 15000 x 384 (config #5)."""
    import torch
    from devqa_amd import lib
    out = {}
    E, L, steps = 8, 2, 6
    g = torch.Generator(device=dev).manual_seed(1)
    w0 = torch.randn((d_out, d_in), device=dev, generator=g) * 0.01
    w = torch.empty((E, d_out, d_in), device=dev)
    mom, var = torch.empty_like(w), torch.empty_like(w)
    dstate = torch.empty((E, L + 1, d_out), device=dev)
    a = torch.rand((E, L, d_in), device=dev, generator=g)
    dy = torch.randn((E, L, d_out), device=dev, generator=g) * 1e-3
    y = torch.empty((E, L, d_out), device=dev)
    one = torch.ones(E, dtype=torch.int32, device=dev)
    t = torch.zeros(E, dtype=torch.int32, device=dev)
    for name, per, fn in (("ft_adamw_step_dense", 16.0 * d_out * d_in, lambda: lib.ft_adamw_step_fm(w, dstate, var, w0, a, dy, y, one, t, 1e-3, 0.9, 0.999, 1e-8, 0.0, -1.0)),
                          ("ft_adamw_step_dense_m_matrix", 24.0 * d_out * d_in, lambda: lib.ft_adamw_step(w, mom, var, w0, a, dy, y, one, t, 1e-3, 0.9, 0.999, 1e-8, 0.0, -1.0))):
        t.zero_()
        for it in range(steps + 2):
            if it == 2:
                torch.cuda.synchronize()
                lib.profile(1)
            t += 1
            fn()
        lib.profile(0)
        ms, _, n = lib.profile_read(lib.PROF_FT_ADAMW)
        out[name] = {"kernel": "ft_adamw_step_kernel (dense [2560,10240] per edit%s)" % ("; first moment from its rank-L factors" if "matrix" not in name else "; first-moment matrix"),
                     "bytes_per_edit_step": per, "edits": E, "launches": int(n), "avg_launch_us": round(1e3 * ms / max(n, 1), 1),
                     "achieved": round(per * E * n / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(per * E * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del w, mom, var
    N, D, Q = 15000, 384, 1000
    corpus = torch.randn((N, D), device=dev, generator=g)
    corpus = corpus / corpus.norm(dim=1, keepdim=True)
    inv = lib.row_inv_norm(corpus)          # cached once per corpus, as the product call sites do (EmbeddingRetriever, IKEvl)
    for q_n in (1, Q):
        queries = torch.randn((q_n, D), device=dev, generator=g)
        for k in (5, 32):
            for it in range(7):
                if it == 2:
                    torch.cuda.synchronize()
                    lib.profile(1)
                lib.cosine_topk(corpus, queries, k, True, True, corpus_inv_norm=inv)
            lib.profile(0)
            ms, wk, n = lib.profile_read(lib.PROF_COSINE)
            us = 1e3 * ms / max(n, 1)
            out["cosine_topk_q%d_k%d" % (q_n, k)] = {
                "kernel": "cosine_fused_kernel (one launch: scores + last-workgroup selection + fp64 re-score)" if q_n <= 4
                          else "score_tile + topk_select (corpus norms cached)", "corpus": [N, D], "queries": q_n, "k": k,
                "bytes_per_query_batch": 4 * N * D, "avg_call_us": round(us, 1), "achieved": round(4.0 * N * D / (us * 1e-6) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(4.0 * N * D / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "gflops": round(2.0 * N * D * q_n / (us * 1e-6) / 1e9, 1)}
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# full-depth parity of the BENCHMARKED configuration (39/12/32 layers): the batched HIP engine vs the oracle cycle that the
# cpu_baseline leg runs anyway.  The oracle (oracle/devqa_oracle.py) is the checker here, never the thing measured.
# ---------------------------------------------------------------------------------------------------------------------------------
POST_ORDER = ["rel", "text_rephrase", "image_rephrase"]


def gpu_parity_capture(editor, cycles_host, n_cmp, E, dev):
    """The first n_cmp of `cycles_host` (devqa_amd.synth.evqa_cycles samples with pre-processed CPU pixel tensors) run as members
    of ONE batch of E cycles through BatchedEditEval -- the benchmarked engine with its shared-prefix packing, column compaction
    and device-side FT loop, at the batch size the timed region uses -- with the label-row logits kept.  -> per compared cycle:
    {pre: {probe: [L, V]}, post: {probe: [L, V]}, steps, losses [steps], accs [12]} (host arrays)."""
    import numpy as np
    import torch
    from devqa_amd.batched import BatchedEditEval, LOC_ORDER, copy_sample

    def on_dev(c):
        def fix(it):
            return dict(it, image=None if it["image"] is None else it["image"].to(dev))
        return {"requests": [fix(c["requests"][0])], "generality": {k: [fix(v[0])] for k, v in c["generality"].items()},
                "locality": {k: [fix(v[0])] for k, v in c["locality"].items()}}
    eds = [on_dev(c) for c in cycles_host[:E]]
    be = BatchedEditEval(editor, cycles_per_batch=len(eds))
    be.keep_debug = True
    out, _ = be.run_batch([copy_sample(c) for c in eds], eds)
    torch.cuda.synchronize()
    caps = []
    for i in range(n_cmp):
        pre, post = {}, {}
        for kind, name, row0, L in be.debug["rows"][i]:
            key = "rel" if kind == "rel" else name
            post[key] = be.debug["post_logits"][row0:row0 + L].float().cpu().numpy()
            if kind == "loc":
                pre[key] = be.debug["pre_logits"][row0:row0 + L].float().cpu().numpy()
        n = int(be.last_steps[i])
        r = out[i]
        accs = [r["reliability"][0]["acc"]] + [r["generality"][k][0]["acc"] for k in ("text_rephrase", "image_rephrase")] + \
               [r["locality"][k][0]["acc"] for k in LOC_ORDER]
        caps.append({"pre": pre, "post": post, "steps": n, "losses": np.array(be.last_losses[i, :n]), "accs": accs, "batch": len(eds)})
    be.debug.clear()
    del be
    torch.cuda.empty_cache()
    return caps


def fp32_parity_capture(cfg, arrays, cycles_host, n_cmp, E, dev, num_layers_last):
    """The same cycles through the same engine in its fp32 ("faithful") compute mode -- exact-fp32 MFMA GEMMs, fp32 weights -- at
    full depth: separates what the algorithmic restructuring (de-duplication, prefix sharing, compaction) changes (nothing, to
    fp32 reassociation) from what bf16 rounding through 39 + 12 + 32 layers changes."""
    import torch
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.synth import IdTokenizer
    model = Blip2Native(cfg, dev, "fp32")
    for n in model._shapes:
        model.load_named_tensors(lambda _n, a=arrays[n]: torch.from_numpy(a), names=[n], refresh=False)
    model.refresh_derived(force=True)
    vllm = BLIP2OPTForEdit(None, dev, model=model, tokenizer=IdTokenizer())
    ft_cfg = FTvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "ft_vl", "blip2-opt-2.7b.yaml"))
    ft_cfg.layers = [num_layers_last]
    caps = gpu_parity_capture(FTvl(vllm, ft_cfg, dev), cycles_host, n_cmp, E, dev)
    del vllm, model
    torch.cuda.empty_cache()
    return caps


def parity_compare(cap, tr, bar):
    """One cycle: engine capture vs oracle trace (faithful_cycle_pretokenized(trace=True)).  Relative errors are |got - ref| over
    the largest |ref| of the probe's label rows (logits, row logsumexp) resp. over max(ref loss, 1) (per-step FT losses) -- the
    scales tests/test_realdim_batched_gpu.py uses.  `decided` rows: the oracle's own top-1 margin exceeds 2 x bar x scale."""
    import numpy as np
    from devqa_amd.batched import LOC_ORDER

    def lse(a):
        m = a.max(1, keepdims=True)
        return (m + np.log(np.exp(a - m).sum(1, keepdims=True)))[:, 0]

    def cmp(got_map, refs, names):
        w = wl = 0.0
        rows = agree = dec_n = dec_ok = 0
        for name, ref in zip(names, refs):
            ref = ref.numpy().astype(np.float64)
            got = got_map[name].astype(np.float64)
            scale = float(np.abs(ref).max())
            w = max(w, float(np.abs(got - ref).max()) / scale)
            wl = max(wl, float(np.abs(lse(got) - lse(ref)).max()) / scale)
            top2 = np.sort(np.partition(ref, -2, axis=1)[:, -2:], axis=1)
            dec = (top2[:, 1] - top2[:, 0]) > 2 * bar * scale
            ok = got.argmax(1) == ref.argmax(1)
            rows += len(ok)
            agree += int(ok.sum())
            dec_n += int(dec.sum())
            dec_ok += int((ok & dec).sum())
        return w, wl, rows, agree, dec_n, dec_ok
    pre = cmp(cap["pre"], tr["rows"][:9], LOC_ORDER)
    post = cmp(cap["post"], tr["rows"][9:], POST_ORDER + LOC_ORDER)
    m = min(cap["steps"], tr["steps"])
    ref_l = np.asarray(tr["losses"][:m], np.float64)
    loss_err = float((np.abs(cap["losses"][:m] - ref_l) / np.maximum(ref_l, 1.0)).max()) if m else 0.0
    return {"steps_gpu": cap["steps"], "steps_ref": tr["steps"], "loss_rel_max": round(loss_err, 5),
            "loss_first_ref": round(float(ref_l[0]), 4) if m else None, "loss_last_ref": round(float(ref_l[-1]), 5) if m else None,
            "logit_rel_max_pre": round(pre[0], 5), "logit_rel_max_post": round(post[0], 5),
            "lse_rel_max": round(max(pre[1], post[1]), 5),
            "argmax_rows_equal": "%d/%d" % (pre[3] + post[3], pre[2] + post[2]),
            "argmax_decided_rows_equal": "%d/%d" % (pre[5] + post[5], pre[4] + post[4]),
            "probes_equal": "%d/12" % sum(abs(a - b) < 1e-6 for a, b in zip(cap["accs"], tr["accs"])),
            "batch_cycles": cap["batch"], "bar": bar}


def parity_block(traces, caps_bf16, caps_fp32):
    out = {"reference": "oracle/devqa_oracle.py faithful_cycle_pretokenized: the reference's call sequence (nothing cached, B = 1), fp32 torch CPU, "
                        "full depth 39/12/32 -- the same cycle cpu_baseline times",
           "quantities": "label-row logits of the 9 pre-edit + 12 post-edit evaluator forwards (error / max |ref| of the probe's rows), their row "
                         "logsumexp, argmax, per-step FT losses (error / max(ref, 1)), executed steps, the 12 per-probe accuracies"}
    if caps_bf16 is not None:
        out["bf16"] = [parity_compare(c, t, 1e-2) for c, t in zip(caps_bf16, traces)]
    if caps_fp32 is not None:
        out["fp32_mode"] = [parity_compare(c, t, 1e-3) for c, t in zip(caps_fp32, traces)]
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# one timed leg (a model + its batches)
# ---------------------------------------------------------------------------------------------------------------------------------
class Leg:
    def __init__(self, args, rank, world, dev, style, layers, keep_host_copy):
        import torch
        from devqa_amd.batched import BatchedEditEval, copy_sample, shard_range
        from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
        from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
        from devqa_amd.synth import IdTokenizer, evqa_cycles, synth_image_u8
        self.args, self.rank, self.world, self.dev = args, rank, world, dev
        t0 = time.time()
        self.model, self.cfg, self.host_arrays = build_full_model(dev, args.seed, layers, threads=build_threads(world),
                                                                  keep_host_copy=keep_host_copy, style=style)
        vllm = BLIP2OPTForEdit(None, dev, model=self.model, tokenizer=IdTokenizer())
        ft_cfg = FTvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "ft_vl", "blip2-opt-2.7b.yaml"))
        ft_cfg.layers = [self.cfg["text_config"]["num_hidden_layers"] - 1]
        self.num_steps = ft_cfg.num_steps
        self.editor = FTvl(vllm, ft_cfg, dev)
        self.be = BatchedEditEval(self.editor, cycles_per_batch=args.cycles_per_step)
        self.build_s = time.time() - t0
        self.vllm = vllm
        self.copy_sample, self.shard_range = copy_sample, shard_range
        self._evqa_cycles, self._synth_image_u8 = evqa_cycles, synth_image_u8
        self._img_cache = {}

    def make_batches(self, K, W, scaling=None, total=None):
        """-> (warm-up batches, timed batches, global id of this rank's first timed cycle, cycles of the timed region on this rank).
        weak: every rank draws its own (W + K) * E cycles.  strong: ONE stream of `total` cycles (the same on every rank), rank r takes
        its contiguous block and cuts it into >= 2 batches of <= E cycles (split_batches)."""
        import torch
        a, E = self.args, self.args.cycles_per_step
        scaling = scaling or a.scaling
        proc, size, vocab = self.vllm.image_processor, self.cfg["vision_config"]["image_size"], self.cfg["text_config"]["vocab_size"]

        def image_of_factory(offset):
            def image_of(s, tag):
                key = (offset + s, tag)
                if key not in self._img_cache:
                    px = torch.from_numpy(proc(self._synth_image_u8(offset + s, tag, size, a.seed)))
                    self._img_cache[key] = px.pin_memory() if getattr(a, "host_pixels", False) else px.to(self.dev)
                return self._img_cache[key]
            return image_of
        if scaling == "weak":
            n = (K + W) * E
            cyc = self._evqa_cycles(n, vocab, a.seed + 7919 * self.rank, image_of_factory(self.rank * n))
            warm, timed, first = cyc[:W * E], cyc[W * E:], self.rank * K * E
        else:
            total = total or K * E
            lo, hi = self.shard_range(total, self.rank, self.world)
            stream = self._evqa_cycles(total, vocab, a.seed, None)       # ids only: images are materialised for this rank's block
            warm_src = self._evqa_cycles(W * E, vocab, a.seed + 104729 * (self.rank + 1), image_of_factory(10 ** 7 + self.rank * W * E))
            img = image_of_factory(0)

            def with_images(s, c):
                def fix(it):
                    return dict(it, image=None if it["image"] is None else img(s, it["image"].split("_")[1]))
                return {"requests": [fix(c["requests"][0])], "generality": {k: [fix(v[0])] for k, v in c["generality"].items()},
                        "locality": {k: [fix(v[0])] for k, v in c["locality"].items()}}
            timed = [with_images(s, stream[s]) for s in range(lo, hi)]
            warm, first = warm_src, lo

        def cut(cs, sizes=None):
            sizes = sizes or [min(E, len(cs) - i) for i in range(0, len(cs), E)]
            out, i = [], 0
            for n_ in sizes:
                out.append(([self.copy_sample(c) for c in cs[i:i + n_]], cs[i:i + n_]))
                i += n_
            return out
        torch.cuda.synchronize()
        return cut(warm), cut(timed, split_batches(len(timed), E) if scaling == "strong" else None), first, timed

    def sample_cycles(self, n):
        """n cycles of this leg's workload (device pixel tensors), for the cpu_baseline leg."""
        import torch
        proc, size = self.vllm.image_processor, self.cfg["vision_config"]["image_size"]

        def image_of(s, tag):
            return torch.from_numpy(proc(self._synth_image_u8(s, tag, size, self.args.seed)))
        return self._evqa_cycles(n, self.cfg["text_config"]["vocab_size"], self.args.seed + 7919 * self.rank, image_of)

    def run(self, batches, pipelined, on_iter=None):
        outs, metas = [], []
        for o, mt in self.be.run_batches(batches, pipelined=pipelined, on_iter=on_iter):
            outs += o
            metas += mt
        return outs, metas

    def reset_stats(self):
        for k in list(self.be.stats):
            self.be.stats[k] = 0

    def close(self):
        import torch
        del self.be, self.editor, self.vllm, self.model
        self._img_cache.clear()
        torch.cuda.empty_cache()


def comm_device(dev):
    """device of the small tensors the collectives move: the GPU under RCCL, the host under gloo (CPU rehearsals of N > 1)"""
    import torch
    import torch.distributed as dist
    return torch.device("cpu") if (dist.is_initialized() and dist.get_backend() == "gloo") else torch.device(dev)


def timed_leg(leg, K, W, barrier, use_dist, rank, world, dev, scaling=None, total=None):
    """W untimed warm-up steps, then the timed region (K steps of E cycles per rank, or this rank's block of the `total`-cycle stream)
    bracketed by barrier + synchronize; max over ranks.  -> (elapsed, n_local, n_total, ranks, gathered score rows on rank 0)"""
    import torch
    import torch.distributed as dist
    from devqa_amd import lib
    from devqa_amd.batched import BatchedEditEval
    from devqa_amd.dist import gather_score_rows
    a = leg.args
    warm, timed, first, _ = leg.make_batches(K, W, scaling, total)
    if warm:
        leg.run(warm, not a.no_pipeline)
    leg.reset_stats()
    barrier()
    # The slot profiler (HIP event pairs around every instrumented launch, on its own stream) is what `roofline` is read from; the pairs cost 1.8 % of
    # throughput when every step carries them (profiles/r03_summary.md U), so they bracket a SAMPLE of the timed region: every --profile-every-th
    # iteration of the pipelined loop (one iteration = stage A of one batch + stage B of the previous one = one step's launches; all steps of the weak
    # leg launch the same shapes).  PROF["frac"] = instrumented share of the region: gemm_roofline / side_kernels scale their per-step figures with it.
    n_it = len(timed)
    P = max(1, int(a.profile_every))
    sample = instrumented_iters(n_it, P, not a.no_pipeline)
    off = bool(os.environ.get("DEVQA_BENCH_NOPROF"))       # A/B of what the event pairs cost: nothing is instrumented
    PROF["frac"], PROF["steps"] = (len(sample) / n_it, sorted(sample)) if sample else (1.0, list(range(n_it)))
    lib.profile(1)            # new recording
    if sample or off:
        lib.profile(0)

    def on_iter(i):
        lib.profile(2 if i in sample else 0)
    t0 = time.time()
    outs, metas = leg.run(timed, not a.no_pipeline, on_iter if (sample and not off) else None)
    barrier()
    elapsed = time.time() - t0
    lib.profile(0)
    n_local = len(outs)
    n_total, ranks = n_local, 1
    got = BatchedEditEval.score_rows(outs, metas, first)
    if use_dist:
        cdev = comm_device(dev)
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nt = torch.tensor([n_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(nt)
        n_total = int(nt.item())
        if (scaling or a.scaling) == "weak":        # equal blocks: the gather's partition rule is shard_range(n_total)
            assert n_total == n_local * world
        got = gather_score_rows(got, n_total, rank, world, cdev)     # the single collective of the data path (RCCL on the GPU)
        ranks = dist.get_world_size()
        if rank == 0:
            assert got.shape == (n_total, 16) and [int(v) for v in got[:, 0]] == list(range(n_total))
    leg.last_batch_sizes = [len(b[1]) for b in timed]
    return elapsed, n_local, n_total, ranks, got


def gemm_roofline(K_E_local, elapsed):
    from devqa_amd import lib
    names = ["gemm_bf16_tn_kernel<32,128,1,4>", "gemm_bf16_glds_kernel<64,128,2,2>", "gemm_bf16_glds_kernel<128,128,2,2>",
             "gemm_bf16_pp_kernel"]
    prof = [lib.profile_read(i) for i in range(4)]
    K_E_local, elapsed = K_E_local * PROF["frac"], elapsed * PROF["frac"]       # the instrumented share of the region
    dom = max(range(4), key=lambda i: prof[i][0])
    g_ms, g_fl, g_n = (sum(p[j] for p in prof) for j in range(3))
    dms, dfl, dn = prof[dom]
    exec_per_cycle = g_fl / max(K_E_local, 1) / 1e12
    alg_scale = min(1.0, A_MIN_TFLOP_PER_CYCLE / exec_per_cycle) if exec_per_cycle > 0 else 0.0
    achieved = (dfl * alg_scale / 1e12) / (dms / 1e3) if dms > 0 else 0.0
    traffic = pmc_traffic(names[dom])
    return {"bound": "mfma", "kernel": names[dom], "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None if traffic is None else traffic["hbm_bytes_per_launch"],
            "traffic_detail": traffic, "avg_launch_us": round(1e3 * dms / max(dn, 1), 2), "launches": int(dn),
            "executed_tflops": round((dfl / 1e12) / (dms / 1e3), 1) if dms > 0 else 0.0,
            "all_gemm_executed_tflops": round((g_fl / 1e12) / (g_ms / 1e3), 1) if g_ms > 0 else 0.0,
            "gemm_time_frac_of_step": round((g_ms / 1e3) / elapsed, 3), "executed_tflop_per_cycle": round(exec_per_cycle, 3),
            "a_min_tflop_per_cycle": A_MIN_TFLOP_PER_CYCLE,
            "instrumented_steps": list(PROF["steps"]), "instrumented_share_of_timed_region": round(PROF["frac"], 4)}


def side_kernels(be, elapsed):
    """HIP-event figures of the other instrumented kernels over the same timed region."""
    from devqa_amd import lib
    frac = PROF["frac"]
    elapsed = elapsed * frac                                  # the instrumented share of the region
    ms, fl, n = lib.profile_read(lib.PROF_ATTENTION)
    att = {"kernel": "attention_mfma_kernel", "launches": int(n), "avg_launch_us": round(1e3 * ms / max(n, 1), 1),
           "tflops_as_launched": round(fl / 1e12 / (ms / 1e3), 1) if ms > 0 else 0.0, "time_frac_of_step": round(ms / 1e3 / elapsed, 3)}
    ms, by, n = lib.profile_read(lib.PROF_LAYERNORM)
    ln = {"kernel": "layernorm_kernel", "launches": int(n), "gbps": round(by / 1e9 / (ms / 1e3), 1) if ms > 0 else 0.0,
          "time_frac_of_step": round(ms / 1e3 / elapsed, 3)}
    ms, _, n = lib.profile_read(lib.PROF_FT_ADAMW)
    by = float(be.stats.get("ft_bytes", 0)) * frac            # (the counter runs over every step; the instrumented steps are a 1-in-N sample of equal batches)
    npad_mean = be.stats.get("npad_sum", 0) / max(be.stats.get("cycles", 1), 1)
    ft = {"bound": "hbm", "kernel": "ft_adamw_step_kernel", "launches": int(n), "avg_launch_us": round(1e3 * ms / max(n, 1), 1),
          "achieved": round(by / 1e9 / (ms / 1e3), 1) if ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": round(by / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4) if ms > 0 else 0.0,
          "algorithmic_bytes": by, "updates": int(be.stats.get("updates", 0)), "npad_mean": round(npad_mean, 1),
          "bytes_rule": "fp32 [2560, npad] per edit: first update 3 tensors (read w0; write w, v), later updates 4 (read + write w, v); the first moment is rebuilt per element from the EMA of dy (devqa_ft_adamw_step_fm), it has no matrix; an edit with ONE loss row has no second-moment matrix either (v = EMA(dy^2) (x) a^2): 2 tensors per update",
          "one_row_edits": int(be.stats.get("one_row_edits", 0)), "one_row_updates": int(be.stats.get("one_row_updates", 0)), "edits": int(be.stats.get("cycles", 0)),
          "time_frac_of_step": round(ms / 1e3 / elapsed, 3)}
    return att, ln, ft


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
                            "write": r["write_bytes_per_launch"], "source": os.path.basename(f) + " (static: separate --pmc passes)"}
        except (OSError, KeyError, ValueError):
            continue
    return None


# ---------------------------------------------------------------------------------------------------------------------------------
def selftest_cpu(args):
    """The multi-rank plumbing without a GPU: gloo process group, barrier-bracketed timed region over fake cycles, MAX all-reduce,
    the all-gather of score rows, rank 0's JSON line.  Not a benchmark."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import devqa_amd  # noqa: F401
    from devqa_amd.batched import shard_range
    from devqa_amd.dist import gather_score_rows, init_from_env
    if "WORLD_SIZE" not in os.environ:      # --gpus 1 without a launcher: a one-rank group
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    rank, world = init_from_env("gloo", min_world=1)
    if os.environ.get("DEVQA_BENCH_FAIL_RANK") == str(rank):
        sys.exit(3)
    E, K = args.cycles_per_step, args.steps
    if args.scaling == "weak":
        first, n_local = rank * K * E, K * E
        sizes = [E] * K
    else:
        lo, hi = shard_range(args.strong_cycles, rank, world)
        first, n_local = lo, hi - lo
        sizes = split_batches(n_local, E)          # the same cut the GPU path makes (Leg.make_batches)
    assert sum(sizes) == n_local and (n_local <= 1 or args.scaling == "weak" or len(sizes) >= 2)
    dist.barrier()
    t0 = time.time()
    rows = np.zeros((n_local, 16), np.float32)
    i = 0
    for n_ in sizes:                               # fake cycles: every score is a fixed function of the GLOBAL cycle id
        ids = np.arange(first + i, first + i + n_)
        rows[i:i + n_, 0] = ids
        for c in range(1, 16):
            rows[i:i + n_, c] = ((ids * 2654435761 + c * 40503) % 1000) / 1000.0
        i += n_
    time.sleep(0.01 * K)
    dist.barrier()
    tt = torch.tensor([time.time() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    nt = torch.tensor([n_local], dtype=torch.int64)
    dist.all_reduce(nt)
    got = gather_score_rows(rows, int(nt.item()), rank, world, torch.device("cpu"))
    if rank == 0 and args.dump_rows:
        np.save(args.dump_rows, got)
    if rank == 0:
        assert [int(v) for v in got[:, 0]] == list(range(int(nt.item())))
        print(json.dumps({"metric": "selftest (not a benchmark)", "value": round(int(nt.item()) / float(tt.item()), 3), "unit": "cycles/s",
                          "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * float(tt.item()) / K, 2),
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "none", "data": "selftest",
                          "rccl_ranks": dist.get_world_size(), "config": {"workload": "fake cycles", "cycles_total": int(nt.item())}}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus and not os.environ.get("DEVQA_FORCE_DIST"):
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%s but --gpus %d; refusing to report a mislabelled run\n" % (env_world, args.gpus))
        sys.exit(2)
    if args.selftest_cpu:
        return selftest_cpu(args)

    import torch
    import torch.distributed as dist
    import devqa_amd  # noqa: F401
    from devqa_amd import lib
    from devqa_amd.dist import init_from_env

    # DEVQA_FORCE_DIST=1: build the process group (RCCL) even for one rank, so that a 1-GPU box walks the N > 1 branches below
    # DEVQA_DIST_BACKEND=gloo + DEVQA_BENCH_SHARE_GPU=1: an N-rank REHEARSAL on fewer GPUs than ranks (collectives over gloo on the host,
    # ranks share the visible devices round-robin; RCCL refuses two ranks on one device) -- walks sharding, sub-batching, both legs and
    # the gather with real cycles on a one-GPU box; not a measurement
    rank, world = init_from_env(backend=os.environ.get("DEVQA_DIST_BACKEND"), min_world=1 if os.environ.get("DEVQA_FORCE_DIST") else 2)
    use_dist = dist.is_initialized()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DEVQA_BENCH_SHARE_GPU"):
        local_rank %= max(torch.cuda.device_count(), 1)
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(dev)
    lib.load()
    layers = None if args.layers is None else tuple(int(x) for x in args.layers.split(","))
    E, K, W = args.cycles_per_step, args.steps, args.warmup
    want_cpu = (not args.no_cpu_baseline) and world == 1

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    head_style = "survey" if args.ffn == "dense" else "opt"
    want_parity = want_cpu and not args.no_parity and layers is None
    n_cmp = max(1, args.cpu_cycles)
    leg = Leg(args, rank, world, dev, head_style, layers, keep_host_copy=want_cpu)
    head_total = args.strong_cycles if args.scaling == "strong" else None
    elapsed, n_local, n_total, ranks, head_rows = timed_leg(leg, K, W, barrier, use_dist, rank, world, dev, args.scaling, head_total)
    head_batches = list(leg.last_batch_sizes)
    out = None
    if rank == 0:
        value = n_total / elapsed
        roof = gemm_roofline(n_local, elapsed)
        # the deduplicated algorithmic minimum is an upper bound on useful work only when the engine executes at least that much;
        # with shared prefixes it executes LESS than SURVEY's A_min on this recipe, so the path fraction uses the smaller figure
        roof["path_frac_of_mfma_peak"] = round(value / world * min(A_MIN_TFLOP_PER_CYCLE, roof["executed_tflop_per_cycle"] or A_MIN_TFLOP_PER_CYCLE)
                                               / MFMA_PEAK_TFLOPS, 4)
        roof["profile_dropped_launches"] = lib.profile_dropped()
        att, ln, ft = side_kernels(leg.be, elapsed)
        steps_mean = leg.be.stats["steps"] / max(leg.be.stats["cycles"], 1)
        d_in = leg.cfg["text_config"]["ffn_dim"]
        out = {
            "metric": "edit+eval cycles/sec, BLIP-2 FT_VL EVQA", "value": round(value, 3), "unit": "cycles/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 2),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic" if not args.host_pixels else "synthetic; MEASUREMENT RUN, not the metric: pixel values cross PCIe inside the timed region",
            "rccl_ranks": ranks,
            "config": {"workload": "BLIP-2-OPT-2.7B + FT_VL, %d synthetic EVQA-shaped edit+eval cycles %s (%d per step), bf16 "
                                   "weights/activations, fp32 master + AdamW state for the edited fc2 matrix, early stop enabled; "
                                   "FFN recipe: %s" % (n_total if args.scaling == "strong" else K * E,
                                                       "in total, block-partitioned over the ranks" if args.scaling == "strong" else "per GPU",
                                                       max(head_batches) if head_batches else E, DENSE_RECIPE if head_style == "survey" else SPARSE_RECIPE),
                       "ffn": "dense" if head_style == "survey" else "sparse", "cycles_per_step": E, "cycles_total": n_total,
                       "mean_ft_steps": round(steps_mean, 2), "ft_active_columns_mean": ft["npad_mean"], "ft_columns": d_in,
                       "layers": "39/12/32" if layers is None else args.layers, "sharding": "splits block-partitioned, 1 gather"},
            "roofline": roof,
            "roofline_hbm": {"ft_adamw_step": ft},
            "kernels": {"attention": att, "layernorm": ln},
            "phase_s": {k: round(v, 3) for k, v in leg.be.stats.items() if k.startswith("t_")},
            "build_s": round(leg.build_s, 1),
        }
        if head_style == "survey":
            out["value_survey_recipe"] = out["value"]
    # ---- the other scaling leg on the same model: both are printed at every N ------------------------------------------------------
    strong_rows = head_rows if args.scaling == "strong" else None
    if not args.no_second_leg:
        other = "strong" if args.scaling == "weak" else "weak"
        o_el, o_local, o_total, _, o_rows = timed_leg(leg, K, 1, barrier, use_dist, rank, world, dev, other,
                                                      args.strong_cycles if other == "strong" else None)
        if other == "strong":
            strong_rows = o_rows
        if rank == 0:
            out[other] = {"scaling": other, "value": round(o_total / o_el, 3), "unit": "cycles/s", "cycles_total": o_total,
                          "elapsed_s": round(o_el, 3), "batches_rank0": list(leg.last_batch_sizes),
                          "note": ("north_star's edit stream: %d cycles block-partitioned over the ranks, every rank cutting its block into >= 2 "
                                   "batches so that the two pipeline stages overlap" % o_total) if other == "strong"
                                  else "every rank runs %d steps of %d cycles" % (K, E)}
    if rank == 0 and args.dump_rows and strong_rows is not None:
        import numpy as np
        np.save(args.dump_rows, strong_rows)
    # the cycles the CPU leg will run (and the batch around them for the parity capture): the head of this rank's own stream
    cycles_for_cpu = leg.sample_cycles(E if want_parity else n_cmp) if want_cpu and rank == 0 else None
    caps = {}
    if want_parity and rank == 0:
        caps["head_bf16"] = gpu_parity_capture(leg.editor, cycles_for_cpu, n_cmp, E, dev)
    host_arrays, cfg = leg.host_arrays, leg.cfg
    last_layer = cfg["text_config"]["num_hidden_layers"] - 1
    leg.close()
    if want_parity and rank == 0:
        caps["head_fp32"] = fp32_parity_capture(cfg, host_arrays, cycles_for_cpu, n_cmp, max(n_cmp, 4), dev, last_layer)
    if rank == 0 and not args.no_hbm_micro and layers is None:
        out["roofline_hbm"].update(hbm_micro(dev))
    # ---- the dense-FFN leg beside the sparse headline ---------------------------------------------------------------------------
    dense_arrays = None
    if args.ffn == "both":
        Kd = args.dense_steps or max(3, K // 3)
        torch.cuda.empty_cache()        # the 22 GB of per-edit AdamW state of this leg should not fight cached blocks of the legs before
        dleg = Leg(args, rank, world, dev, "survey", layers, keep_host_copy=want_parity)
        d_elapsed, d_local, d_total, _, _ = timed_leg(dleg, Kd, 2, barrier, use_dist, rank, world, dev, "weak")
        if rank == 0:
            _, _, dft = side_kernels(dleg.be, d_elapsed)
            out["dense_ffn"] = {"value": round(d_total / d_elapsed, 3), "unit": "cycles/s", "steps": Kd, "warmup": 2,
                                "ms_per_step": round(1e3 * d_elapsed / Kd, 2), "recipe": DENSE_RECIPE,
                                "mean_ft_steps": round(dleg.be.stats["steps"] / max(dleg.be.stats["cycles"], 1), 2),
                                "ft_active_columns_mean": dft["npad_mean"], "ft_adamw_step": dft,
                                "phase_s": {k: round(v, 3) for k, v in dleg.be.stats.items() if k.startswith("t_")}}
            out["value_survey_recipe"] = out["dense_ffn"]["value"]      # SURVEY 8(d)'s weight recipe verbatim: the contract figure
            if want_parity:
                caps["dense_bf16"] = gpu_parity_capture(dleg.editor, cycles_for_cpu, n_cmp, E, dev)
        dense_arrays = dleg.host_arrays
        dleg.close()
    # ---- BASELINE configs #3-#5 on this GPU (bounded; one rank only) ------------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_configs and layers is None:
        out["configs"] = run_config_legs(args)
    if rank == 0:
        if want_cpu and cycles_for_cpu is not None:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 64)
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(cfg, args.seed, threads, host_arrays, cycles_for_cpu, n_cmp, head_style, trace=want_parity)
            if want_parity:
                out["parity"] = parity_block(cpu_baseline.last_traces, caps.get("head_bf16"), caps.get("head_fp32"))
                out["parity"]["recipe"] = "dense (survey)" if head_style == "survey" else "sparse (opt)"
                host_arrays = None
                if dense_arrays is not None and "dense_bf16" in caps:
                    # parity only (not a timing): distinct images encoded once -- same values, a third of the CPU time
                    dcpu = cpu_baseline(cfg, args.seed, threads, dense_arrays, cycles_for_cpu, n_cmp, "survey", trace=True, cache_images=True)
                    out["dense_ffn"]["parity"] = parity_block(cpu_baseline.last_traces, caps["dense_bf16"], None)
                    out["dense_ffn"]["parity"]["cpu_seconds_per_cycle"] = dcpu["seconds_per_cycle"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        from devqa_amd.dist import close_score_comms
        dist.barrier()
        torch.cuda.synchronize()
        close_score_comms()             # the library's own RCCL communicator goes before torch's process group
        dist.destroy_process_group()


def run_config_legs(args):
    """BASELINE.json configs #3, #4, #5 at full model dims on this GPU, a bounded number of cycles each (tools/bench_configs.py):
    cycles/s, the GEMM family's roofline fraction from the slot profiler, the algorithmic TFLOP/cycle used.  A leg that fails
    reports its error instead of ending the run."""
    import gc
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc
    n = [int(x) for x in args.config_cycles.split(",")]
    legs = [("llava_ft_vl", lambda: bc.llava_ft(n[0], 16)), ("blip2_mend_vl", lambda: bc.blip2_mend(n[1])),
            ("minigpt4_ike_vl", lambda: bc.minigpt4_ike(n[2]))]
    out = {}
    cwd = os.getcwd()
    import contextlib
    for name, fn in legs:
        try:
            with contextlib.redirect_stdout(sys.stderr):     # the evaluator's progress prints: stdout carries the JSON line only
                out[name] = fn()
        except Exception as e:      # reported, never silent
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        os.chdir(cwd)
        gc.collect()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
