#!/usr/bin/env python3
"""bench.py -- dialog-rounds/sec of the enc_dec_a train step on MI355X (BASELINE.json metric).

One "step" = one full training step of the ViLBERT-dialog encoder-decoder on a synthetic batch that is
already resident in HBM: dropout-on forward + LM cross-entropy + backward + (N>1: RCCL gradient all-reduce,
overlapped with backward) + fused AdamW update, bf16 storage / fp32 accumulation.  Nothing is skipped.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python bench.py --gpus N --steps K --warmup W          # N > 1 without RANK/WORLD_SIZE: bench.py starts its own N ranks
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          # ... or is started as one of them

`--gpus N` is a promise about the run, not a hint: the line says n_gpus = N and config.rccl.ranks_seen_by_allreduce = N or the
process exits non-zero.  A plain `python bench.py --gpus N` (no rank environment) is the LAUNCHER: before any GPU call it checks
that N devices are visible and starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (never
an exec), relays rank 0's JSON as its own last line and propagates the exit code (train_gen.py:295,324-329 -- the reference's
DataParallel wrapping -- is what the N ranks replace).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel (bf16 MFMA GEMM family): algorithmic FLOPs per launch / average launch
                duration measured with HIP events on the launch stream, vs the 2.5 PFLOP/s dense bf16 peak
  cpu_baseline  the CPU oracle (a parity-checked restatement of the reference, oracle/vd_oracle.py) timed on the
                host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_ROW_TRAIN = 274.0e9      # useful fwd 91.4 GF + bwd 182.6 GF per dialog round, T=256,R=37,U=25
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0             # HBM3E peak, same guide (6.3 TB/s is what a float4 copy reaches)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def parse_last_json(text):
    """The contract's "last stdout line is the JSON": returns (object or None, the text in front of that line)."""
    lines = text.rstrip("\n").split("\n")
    for i in range(len(lines) - 1, -1, -1):
        ln = lines[i].strip()
        if not ln:
            continue
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln), "\n".join(lines[:i])
            except ValueError:
                return None, text
        return None, text
    return None, text


def check_rank_count(line, n):
    """None if the JSON line really describes an n-rank run, else the reason it does not."""
    if not isinstance(line, dict):
        return "rank 0 printed no JSON line"
    if line.get("n_gpus") != n:
        return "the line says n_gpus = %r, asked for %d" % (line.get("n_gpus"), n)
    rccl = (line.get("config") or {}).get("rccl") or {}
    if n > 1 and (rccl.get("world_size") != n or rccl.get("ranks_seen_by_allreduce") != n):
        return ("the communicator spans world_size = %r / ranks_seen_by_allreduce = %r ranks, asked for %d"
                % (rccl.get("world_size"), rccl.get("ranks_seen_by_allreduce"), n))
    return None


def launch_ranks(n, argv, worker_cmd=None, visible_gpus=None, env=None, timeout=None):
    """`python bench.py --gpus n` without a rank environment: start the n ranks as a child process group and relay rank 0's line.

    Runs BEFORE anything touches the GPU (torch.cuda.device_count() does not initialise it on this stack; is_available() would)
    and never replaces this process: `python -m torch.distributed.run` is a child (subprocess), its exit code is returned.
    Returns (exit code, JSON object or None, relayed stdout in front of the JSON line).
      worker_cmd     what each rank runs (default: this file); a test passes a stub that needs no GPU
      visible_gpus   override of the device count (tests)"""
    import subprocess
    one_gpu_validation = bool((env or os.environ).get("GSTVD_BENCH_ONE_GPU"))
    if visible_gpus is None:
        visible_gpus = torch.cuda.device_count()
    if visible_gpus < n and not one_gpu_validation:
        sys.stderr.write("bench: --gpus %d asked for, %d GPU(s) visible on this node: refusing to report a %d-GPU number "
                         "from fewer devices (run with --gpus %d, or on a node with %d GPUs)\n"
                         % (n, visible_gpus, n, max(visible_gpus, 1), n))
        return 4, None, ""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += list(worker_cmd) if worker_cmd else [os.path.abspath(__file__)]
    cmd += list(argv)
    cenv = dict(env if env is not None else os.environ)
    cenv.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this host driver
    cenv.setdefault("OMP_NUM_THREADS", "8")
    sys.stderr.write("bench: launching %d ranks: %s\n" % (n, " ".join(cmd)))
    try:
        r = subprocess.run(cmd, env=cenv, stdout=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        sys.stderr.write("bench: the %d-rank run did not finish within %s s\n" % (n, timeout))
        return 5, None, ""
    text = r.stdout.decode("utf-8", "replace")
    line, head = parse_last_json(text)
    rc = r.returncode
    if rc == 0:
        why = check_rank_count(line, n)
        if why:
            sys.stderr.write("bench: the %d-rank run is not a valid %d-GPU measurement: %s\n" % (n, n, why))
            rc = 6
    else:
        sys.stderr.write("bench: the %d-rank run exited with code %d\n" % (n, rc))
    return rc, (line if rc == 0 else None), (head if line is not None else text)


def launcher_main(n, argv):
    rc, line, head = launch_ranks(n, argv)
    if head.strip():
        print(head, flush=True)
    if line is not None:
        line.setdefault("config", {})["launched_by"] = "bench.py --gpus %d (self-launched %d ranks via torch.distributed.run, child process)" % (n, n)
        print(json.dumps(line), flush=True)
    return rc


def run_leg(name, extra, base_argv, rank, world, local, port, timeout=420):
    """One extra measurement beside the headline (a different rows/GPU or gradient payload), each in a FRESH child process per
    rank (own process group on its own port, own hipGraph capture): a failure there costs that leg, never the headline line.
    Every rank calls this with the same arguments after the parent's process group is gone; rank 0 returns the leg's JSON."""
    import subprocess
    # (without the elastic agent's variables: TORCHELASTIC_USE_AGENT_STORE would make rank 0 of the leg look for the PARENT
    # launcher's store on the leg's port instead of opening its own -- the leg's rendezvous then waits for ever)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, os.path.abspath(__file__)] + list(base_argv) + list(extra) + ["--leg", name]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "leg %s: no result within %d s" % (name, timeout)}
    if rank != 0:       # (every rank of a leg exits with the same code: its ranks agree on success through a collective before they time)
        return None if r.returncode == 0 else {"error": "leg %s: child exit code %d" % (name, r.returncode)}
    line, _ = parse_last_json(r.stdout.decode("utf-8", "replace"))
    if r.returncode != 0 or line is None:
        return {"error": "leg %s: child exit code %d%s" % (name, r.returncode, "" if line else ", no JSON line")}
    return line


def synthetic_rows(B, T, R, U, F, V, seed, device):
    """tensors as train_gen.forward hands them to the model (post row-sampling)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, min(30000, V), (B, T), generator=g)
    lens = torch.randint(int(0.6 * T), T + 1, (B,), generator=g)
    pos = torch.arange(T)[None]
    ids[:, 0] = 101
    utt = torch.randint(6, 14, (B, T), generator=g).cumsum(1)          # utterance boundaries
    sep = torch.zeros(B, T, dtype=torch.bool)
    sep.scatter_(1, utt.clamp(max=T - 1), True)
    ids[sep] = 102
    seg = (sep.long().cumsum(1) - sep.long() + 1) % 2
    keep = pos < lens[:, None]
    ids = ids * keep
    seg = seg * keep
    att = (ids != 0).float()
    feats = torch.randn(B, R, F, generator=g).abs()
    feats[:, 0] = feats[:, 1:].mean(1)
    loc = torch.rand(B, R, 5, generator=g)
    loc[:, 0] = torch.tensor([0., 0., 1., 1., 1.])
    img_mask = torch.ones(B, R)
    alen = torch.randint(3, 11, (B,), generator=g)
    ans = torch.randint(1000, min(30000, V), (B, U), generator=g)
    upos = torch.arange(U)[None]
    dec_ids = torch.zeros(B, U, dtype=torch.long)
    dec_ids[:, 1:] = (ans * (upos < alen[:, None]))[:, :-1]
    dec_ids[:, 0] = 101
    labels = ans * (upos < alen[:, None])
    labels[torch.arange(B), alen] = 102
    dec_att = (upos < (alen[:, None] + 2)).float()
    d = dict(enc_image_features=feats, enc_image_spatials=loc, enc_image_mask=img_mask, enc_input_ids=ids,
             enc_segments=seg, enc_attention_mask=att, dec_input_ids=dec_ids, dec_attention_mask=dec_att, dec_labels=labels)
    return {k: v.to(device) for k, v in d.items()}


def build_model(device, precision, seed, streams=True):
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    from gst_visdial_amd.modules import VisualDialogEncoder, VisualDialogDecoder, EncoderDecoderModel
    d = tempfile.mkdtemp(prefix="gstvd_bench_")
    with open(os.path.join(d, "enc.json"), "w") as f:
        json.dump(bert_base_enc_config(), f)
    with open(os.path.join(d, "dec.json"), "w") as f:
        json.dump(bert_base_dec_config(), f)
    params = dict(model_enc_config=os.path.join(d, "enc.json"), model_dec_config=os.path.join(d, "dec.json"), gpu_ids=[0],
                  model="enc_dec_a", mode="vd_train", batch_size=16, device=device, amd_precision=precision, amd_seed=seed,
                  amd_streams=streams)
    torch.manual_seed(seed)
    enc, dec = VisualDialogEncoder(params), VisualDialogDecoder(params)
    model = EncoderDecoderModel(params, enc, dec)
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings        # train_gen.py:293
    return model.to(device), params


def cpu_baseline(model, T, R, U, F, V, rows=2):
    """The oracle (parity-checked port of the reference) on the host cores: forward + loss + backward, fp32, TRAIN mode.

    BASELINE.md section 4's procedure: model.train() (the oracle's `train=True`: F.dropout on the host RNG at every one of the
    reference's dropout sites -- 11 % of the reference's own CPU time is bernoulli_), 1 untimed warm-up + 5 timed steps of the
    same sample (round 6: 5 instead of 3 -- the only CPU figure of the line swung 20 % between rounds), MEDIAN reported with the
    min-max spread, thread count and logical-CPU count printed.  One deviation, stated in the line: the thread count
    is min(os.cpu_count(), 16), not os.cpu_count() -- eager fp32 PyTorch on the 256-thread GPU-box host collapses with every
    logical CPU in the pool (measured 900 s for 2 rows with 256 threads)."""
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    from oracle import vd_oracle as O
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    keys = [k for k in O.live_param_keys(sd) if k in sd]
    enc_cfg, dec_cfg = bert_base_enc_config(), bert_base_dec_config()

    def one(B, Tt):
        b = synthetic_rows(B, Tt, R, U, F, V, 999, "cpu")
        t0 = time.perf_counter()
        O.grads(sd, enc_cfg, dec_cfg, b, keys, wrt_feats=False, train=True)
        return time.perf_counter() - t0

    one(1, 16)                                     # thread-pool / allocator warm-up (tiny, untimed)
    t1 = one(1, T)                                 # sizing probe: how many rows fit ~4 s per timed step (so 1 + 5 steps stay ~25 s)
    n = max(1, min(32, int(4.0 / max(t1, 1e-3))))
    one(n, T)                                      # the procedure's warm-up step (untimed)
    times = sorted(one(n, T) for _ in range(5))
    dt = times[2]                                  # median of 5
    spread = (times[-1] - times[0]) / dt
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": n / dt, "unit": "dialog-rounds/sec", "cores": threads, "kind": "port", "cpu_model": cpu_model,
            "host_logical_cpus": os.cpu_count(), "mode": "train (dropout on, host RNG)", "timed_steps": 5, "warmup_steps": 1,
            "step_seconds": [round(t, 2) for t in times], "spread_of_median": round(spread, 3),
            "sample": "%d row(s) x (1 warm-up + 5 timed) train steps (fwd+loss+bwd, fp32, dropout on, T=%d R=%d U=%d) of oracle/vd_oracle.py "
                      "on %d of %d logical CPUs (eager fp32 PyTorch collapses with all of them in the pool); median %.1f s per step, "
                      "min-max spread of the 5 steps %.0f %% of the median" % (n, T, R, U, threads, os.cpu_count() or 0, dt, 100 * spread)}


def eval_decode_side(device, V):
    """BASELINE configs[3] beside the headline (outside the timed region): the sampling decode of generate.py /
    visual_dialog_model.py:86-111 (16 rows x 18 token steps, KV cache, the token loop replayed as one hipGraph) and the
    evaluate_gen.py:76-106 candidate scoring (one 500-row chunk = 5 rounds x 100 options, encode-once vs the reference's expanded
    batch), bf16, full-size model, synthetic inputs resident in HBM."""
    model, params = build_model(device, "bf16", seed=1)
    model.eval()

    def timeit(fn, n=5, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    out = {}
    with torch.no_grad():
        params["mode"] = "vd_eval_val"
        R_, O_ = 5, 100
        b = synthetic_rows(R_, 256, 37, 25, 2048, V, 5, device)
        cand = synthetic_rows(R_ * O_, 256, 37, 25, 2048, V, 6, device)
        enc = [b[k] for k in ("enc_image_features", "enc_image_spatials", "enc_image_mask", "enc_input_ids", "enc_segments", "enc_attention_mask")]

        def once():
            return model.score_candidates(*enc, cand["dec_input_ids"].clone(), cand["dec_attention_mask"], O_)

        def expanded():
            rep = lambda x: x.repeat_interleave(O_, 0)          # noqa: E731
            return model(enc_image_features=rep(enc[0]), enc_image_spatials=rep(enc[1]), enc_image_mask=rep(enc[2]), enc_input_ids=rep(enc[3]),
                         enc_segments=rep(enc[4]), enc_attention_mask=rep(enc[5]), dec_input_ids=cand["dec_input_ids"].clone(),
                         dec_attention_mask=cand["dec_attention_mask"], dec_labels=None, loss_reduction=False)
        t_once = timeit(once)
        t_exp = timeit(expanded, n=2, warm=1)
        out["score"] = {"chunk": "5 rounds x 100 options = 500 candidate rows (evaluate_gen.py:29,151)", "encode_once_ms": round(t_once, 2),
                        "expanded_batch_ms": round(t_exp, 2), "candidates_per_s": round(500e3 / t_once, 1)}
        params["mode"] = "vd_gen_val"
        d = synthetic_rows(16, 256, 37, 25, 2048, V, 7, device)
        kw = {k: d[k] for k in ("enc_image_features", "enc_image_spatials", "enc_image_mask", "enc_input_ids", "enc_segments", "enc_attention_mask")}
        kw.update(dec_input_ids=torch.full((16, 1), 101, dtype=torch.long, device=device), temperature=0.7, top_k=7, top_p=0.0,
                  ngram_blocking_size=0)
        params["amd_decode_graph"] = True
        t_rep = timeit(lambda: model(**kw))
        kw["ngram_blocking_size"] = 4               # the questioner's setting (generate.py:138-141)
        t_ng = timeit(lambda: model(**kw))
        lpt = getattr(model.engine, "decode_lib_calls_per_token", None)       # counted on the eager first call (each >= 1 launch)
        out["decode"] = {"rows": 16, "steps": 18, "ms_replayed": round(t_rep, 2), "ms_replayed_ngram4": round(t_ng, 2),
                         "ms_per_token": round(t_rep / 18, 3), "launches_per_token": round(lpt, 1) if lpt else None, "rows_per_s": round(16e3 / t_rep, 1),
                         "how": "encoder + cross-K/V once, KV-cached decoder, sampling step on the device, token loop = one hipGraph replay"}
    del model
    torch.cuda.empty_cache()
    return out


def demangle(sym):
    """Best effort (llvm-cxxfilt from the ROCm tree / c++filt); the mangled symbol is what rocprofv3 traces key on."""
    import shutil
    import subprocess
    for exe in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", shutil.which("c++filt")):
        if exe and os.path.exists(exe):
            try:
                return subprocess.run([exe, sym], stdout=subprocess.PIPE, timeout=10).stdout.decode().strip()
            except Exception:       # noqa: BLE001
                pass
    return None


def _flush_c_stdio():
    """RCCL (NCCL_DEBUG=VERSION) prints through C stdio; flush it so that bench.py's JSON is the last stdout line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows-per-gpu", type=int, default=16)
    ap.add_argument("--seq-len", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--grad-compress", default="auto", choices=["auto", "none", "bf16"],
                    help="dtype of the gradient all-reduce payload (auto: bf16 for N>1 -- halves xGMI traffic; fp32 master grads kept)")
    ap.add_argument("--breakdown-json", default=None, help="write the per-kernel breakdown here")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off", "segmented"],
                    help="replay the whole train step as one hipGraph (auto = on at every N: at N>1 the RCCL all-reduces are captured "
                         "with the step and a failed capture is a hard error; off: eager issue, host bound)")
    ap.add_argument("--no-pipeline", action="store_true", help="diagnostic: no backward pipeline (wgrad/AdamW after backward, same stream)")
    ap.add_argument("--chunk-melems", type=int, default=0,
                    help="backward-pipeline slice size in Mi elements (0 = auto: 192 at N=1; graded 22,27,27,27,27,96,96,32,16 at N>1)")
    ap.add_argument("--chunk-list", default="", help="graded backward-pipeline slice sizes in Mi elements, comma separated (overrides --chunk-melems)")
    ap.add_argument("--no-streams", action="store_true", help="run the vision stream on the main HIP stream")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode timing beside the bf16 headline")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive measurement (inputs staged from pinned host memory)")
    ap.add_argument("--no-eval-decode", action="store_true", help="skip the BASELINE configs[3] side measurements (decode / candidate scoring)")
    ap.add_argument("--shard-update", default="off", choices=["on", "off"],
                    help="N>1: reduce-scatter the gradient slices, AdamW on the rank's 1/N shard, all-gather the bf16 shadow weights "
                         "(pipeline.BackwardPipeline(shard_update=True)) instead of all-reduce + the full AdamW on every rank")
    ap.add_argument("--legs", default="auto", choices=["auto", "on", "off"],
                    help="extra N>1 legs beside the headline, each in fresh child processes: 10 rows/rank (BASELINE configs[2]) and the "
                         "reference-faithful fp32 gradient all-reduce (auto: when more than one rank runs)")
    ap.add_argument("--no-rows-sensitivity", action="store_true",
                    help="skip the 32 / 64 rows-per-GPU side runs (fresh child processes; never `value`) that separate 'the kernels cannot' "
                         "from '16 rows per GPU is too small' for north_star's co-attention target")
    ap.add_argument("--leg", default=None, help=argparse.SUPPRESS)          # set by run_leg(): this process IS a leg's rank
    args = ap.parse_args()
    argv = sys.argv[1:]

    if args.leg is not None:             # a leg's rank: the timed region only, no side measurements, no legs of its own
        keep_breakdown = args.leg.startswith("rows_sens")       # (the rows-sensitivity children exist FOR the per-kernel breakdown)
        args.no_breakdown = args.no_breakdown or not keep_breakdown
        args.no_cpu_baseline = args.no_h2d = args.no_fp32 = args.no_eval_decode = args.no_rows_sensitivity = True
        args.legs = "off"

    # ---- launcher: `--gpus N` with N > 1 and no rank environment -> start the N ranks ourselves (child process, no GPU call here)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launcher_main(args.gpus, argv))

    # torch's OpenMP pool on a 256-logical-CPU host: workers that keep spinning after a parallel CPU tensor op starve the ROCm
    # runtime's threads and slow hipGraph replays (measured 2x); the GPU legs need no CPU parallelism,
    # the CPU baseline sets its own thread count later
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    leg_ports = []

    def leg_port(k):
        """Rendezvous port of the k-th child leg.  With a process group, rank 0 asked the kernel for free ports right after the
        group came up and broadcast them (`leg_ports`): the neighbours of MASTER_PORT may be in use, or shared with a second
        bench on the same host.  Without one (a leg of a 1-rank run) any free port will do."""
        if k < len(leg_ports):
            return leg_ports[k]
        if world == 1:
            return _free_port()
        return int(os.environ.get("MASTER_PORT", "29511")) + 1 + k

    if world != args.gpus:
        # a --gpus N line from a different number of ranks would be a mislabelled measurement: never print one
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE = %d: start it as `python bench.py --gpus %d` (self-launching) or under "
                         "torch.distributed.run --nproc-per-node %d\n" % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # GSTVD_BENCH_ONE_GPU=1 (validation only): run the multi-rank control flow with every rank on cuda:0 and the collectives
    # over gloo (RCCL refuses two ranks on one device; gloo cannot be graph-captured, so the step is issued eagerly)
    one_gpu = bool(os.environ.get("GSTVD_BENCH_ONE_GPU")) and world > 1
    if one_gpu:
        local = 0
        if args.graph == "auto":
            args.graph = "off"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if rank != 0:                             # only rank 0 reports; keep the other ranks' C-level chatter off the shared stdout
        try:
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        except OSError:
            pass
    force_dist = bool(os.environ.get("GSTVD_FORCE_DIST"))      # exercise the RCCL path on one GPU (validation only)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            from gst_visdial_amd.graph import enable_watchdog_introspection
            enable_watchdog_introspection()      # lets the capture wait until c10d's watchdog has retired the warm-up collectives
            dist.init_process_group("nccl", device_id=device)
        # ports for the child legs, agreed on while the group is young and healthy (nothing is issued on it after a failed capture)
        ports = [[_free_port() for _ in range(6)] if rank == 0 else None]
        dist.broadcast_object_list(ports, src=0)
        leg_ports.extend(int(p) for p in ports[0])

    from gst_visdial_amd import ops
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline

    B, T, R, U, F = args.rows_per_gpu, args.seq_len, 37, 25, 2048
    model, params = build_model(device, args.precision, seed=1234, streams=not args.no_streams)      # same init on every rank
    V = model.decoder.config.vocab_size
    model.train()
    batch = synthetic_rows(B, T, R, U, F, V, 1234 + rank, device)
    opt = FusedAdamW(model, lr=2e-5, warmup_steps=1500, t_total=100000)
    # gradients are finalised slice by slice on a third stream during backward: grouped wgrad GEMMs -> column
    # reductions -> (N>1) RCCL all-reduce of the slice -> fused AdamW on the slice
    compress = {"auto": "bf16" if (world > 1 or force_dist) else None, "none": None, "bf16": "bf16"}[args.grad_compress]
    # Slice size: at N>1 many slices let each all-reduce overlap the rest of backward and keep the exposed tail (last
    # slice's wgrad + all-reduce + AdamW) short; at N=1 there is nothing to hide and two large slices are faster (measured
    # 15.4 ms at 40 Mi, 14.8 ms at 192 Mi: the grouped wgrad launches are larger, AdamW streams less often through L2)
    if args.chunk_list:
        chunk_elems = [int(c) << 20 for c in args.chunk_list.split(",")]
    elif args.chunk_melems:
        chunk_elems = args.chunk_melems << 20
    elif world > 1 or force_dist:
        # graded: the decoder + LM head (first to finish) go out as one large slice, the encoder's follow in shrinking
        # ones so that the exposed tail after backward (last slice's wgrad + all-reduce + AdamW) stays short
        # (tools/dist_slice_sweep.sh on the 1-rank RCCL path: 14.28 / 12.37 ms at 16 / 10 rows against 14.80 / 12.81 ms for the
        # six-slice list 128, 48, 48, 32, 32, 24 of round 1 -- fewer, larger grouped launches; the last two slices stay small)
        # round 3: the decoder's part goes out in five small slices DURING its own backward (LM head, then three decoder layers
        # at a time) instead of one 128 Mi slice at its end -- the first all-reduce starts ~2 ms earlier and the weight-gradient /
        # AdamW work of those slices runs beside a latency-bound chain: 14.04 vs 14.28 ms on the 1-rank RCCL path
        # (profiles/r03_chunk_sweep.txt); messages stay >= 44 MB (bf16), large enough for a point-to-point xGMI ring
        chunk_elems = [c << 20 for c in (22, 27, 27, 27, 27, 96, 96, 32, 16)]
    else:
        # N = 1: ONE slice after backward.  Its grouped weight-gradient launch applies AdamW to its weights in the tiles' epilogues
        # (gstvd_gemm_grouped_adamw, pipeline.py `fuse_update`) -- the HBM-bound update streams under the other workgroups' K-loops
        # -- and a short remainder pass covers biases / LayerNorm / embeddings: 12.82 ms against 13.23-13.30 ms for the two
        # launches and 13.35-13.39 ms for round 3's list (22, 27, 27, 27, 27, 192 Mi: weight gradients + AdamW of the decoder's
        # slices beside its backward chain), same box, profiles/r04_fuse_check.txt.  Slices during backward no longer pay at N = 1:
        # beside the encoder's backward they only move time around, and a slice's AdamW takes the chip's wave slots from the
        # decoder's chain (its next kernel waits for the whole update: tools/r04_bg_sweep.sh, profiles/r04_bg_sweep.txt).
        chunk_elems = 1 << 40
    pipe = None if args.no_pipeline else BackwardPipeline(model.engine, optimizer=opt, chunk_elems=chunk_elems,
                                                          compress=compress, force_collective=force_dist,
                                                          shard_update=(args.shard_update == "on" and (world > 1 or force_dist)),
                                                          direct_bf16=os.environ.get("GSTVD_BENCH_DIRECT_BF16", "1") != "0")

    def device_step():
        """Everything of a train step that is device work -- the part that is captured into the hipGraph."""
        loss, _ = model(**batch)
        loss.backward()          # includes the pipelined all-reduce + AdamW
        opt.step()               # no-op marker: the update was applied during backward
        opt.zero_grad()
        return loss

    def host_step_end():
        """Host side of train_gen.py:329 (scheduler.step()): stays OUTSIDE the captured function, or it would run once at
        capture time and never again -- the schedule position advances every step and the device learning-rate table is
        refreshed (a 2 KB pinned-memory copy) whenever the value changed."""
        opt.scheduler_step()
        opt.upload_lr()

    def step():
        loss = device_step()
        host_step_end()
        return loss

    def barrier():
        if world > 1 or force_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    eager_step = step
    capture_error = None
    # hipGraph replay is the default at every N: ~1000 launches cost ~20 ms of host time per step when issued from Python,
    # more than the GPU needs to execute them.  At N>1 the RCCL all-reduces of the backward pipeline are captured with the
    # rest of the step (stream capture of RCCL collectives, as used for graph-mode serving on this stack); if capture is
    # refused the run falls back to eager issue and says so in config.hip_graph.
    use_graph = args.graph in ("on", "auto", "segmented")
    # `--graph segmented` (or GSTVD_FORCE_SEGMENTED=1 with the default `auto`): the N > 1 fall-back form -- one hipGraph per gradient
    # slice, the slices' collectives and updates issued eagerly between the replays (graph.SegmentedStep)
    segmented = (world > 1 or force_dist) and pipe is not None and (args.graph == "segmented" or (args.graph == "auto" and os.environ.get("GSTVD_FORCE_SEGMENTED", "0") == "1"))
    if args.graph == "segmented" and not segmented:
        sys.stderr.write("bench: --graph segmented needs the N > 1 path (a collective): world %d, GSTVD_FORCE_DIST %s\n" % (world, force_dist))
        sys.exit(2)
    for _ in range(max(args.warmup, 2) if use_graph else args.warmup):
        loss = step()
    if use_graph and segmented:
        from gst_visdial_amd.graph import SegmentedStep
        try:
            replay = SegmentedStep(device_step, pipe, warmup=0)
        except Exception as ex:          # noqa: BLE001
            capture_error = "%s: %s" % (type(ex).__name__, ex)
            torch.cuda.synchronize()
        import torch.distributed as dist
        ok = torch.tensor([0.0 if capture_error else 1.0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            sys.stderr.write("bench: segmented capture of the train step failed on %s (%s)\n" % ("this rank" if capture_error else "another rank", capture_error))
            dist.destroy_process_group()
            sys.exit(3)

        def step():                  # noqa: F811
            loss = replay()
            host_step_end()
            return loss
        use_graph = "segmented"
        for _ in range(args.warmup):
            loss = step()
    elif use_graph:
        # Capture one full step (forward, backward on both HIP streams, pipelined all-reduce + fused AdamW) and replay it:
        # same kernels, same work, no host in the loop.  Device-resident state (dropout offset, AdamW step counter)
        # advances inside the graph.
        from gst_visdial_amd.graph import GraphedStep
        try:
            replay = GraphedStep(device_step, warmup=0)
        except Exception as ex:          # noqa: BLE001
            capture_error = "%s: %s" % (type(ex).__name__, ex)
            torch.cuda.synchronize()
        if world > 1 or force_dist:
            # every rank must take the same decision: a rank that replays while another issues eagerly would mismatch
            # collective counts and hang.  And at N>1 the eager path is host bound (~26 ms of launches for ~15 ms of GPU work),
            # so a silent fall-back would report a throughput that is not the product's: fail loudly instead.
            import torch.distributed as dist
            ok = torch.tensor([0.0 if capture_error else 1.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                # every rank takes the same decision.  `--graph on`: the caller asked for the replayed step, refuse anything else.
                # `--graph auto` (the default): time the eager path on ALL ranks and say so -- a host-bound but valid number
                # (config.hip_graph = false, config.capture_error) is worth more to a scaling run than no number at all.
                sys.stderr.write("bench: hipGraph capture of the train step failed on %s (%s)\n"
                                 % ("this rank" if capture_error else "another rank", capture_error))
                if args.graph == "on" or args.leg is not None:
                    sys.stderr.write("bench: --graph on: refusing to time another path\n")
                    dist.destroy_process_group()
                    sys.exit(3)
                capture_error = capture_error or "capture failed on another rank"
                # An invalidated capture can leave the communicator (and c10d's watchdog) in an undefined state: nothing more is
                # issued on this process group.  The labelled eager measurement runs in FRESH child processes (own group, own
                # port, --graph off), one per rank; rank 0 relays the child's line with the capture error attached.
                try:
                    dist.destroy_process_group()
                except Exception:          # noqa: BLE001
                    pass
                # first the SEGMENTED form (one graph per gradient slice, collectives eager between them: the same kernels at ~40 host
                # calls per step), then -- if that capture is refused as well -- eager issue (host bound, ~19 ms per step)
                fb = run_leg("segmented_fallback", ["--graph", "segmented"], argv, rank, world, local, leg_port(0))
                how = "fresh child processes after the failed whole-step hipGraph capture: SEGMENTED replay (graph.SegmentedStep)"
                if fb is not None and "error" in fb:       # (rank > 0: None = its child exited 0; the ranks of a leg fail together)
                    if rank == 0:
                        sys.stderr.write("bench: the segmented fall-back failed too (%s); timing eager issue\n" % fb.get("error"))
                    fb = run_leg("eager_fallback", ["--graph", "off"], argv, rank, world, local, leg_port(5))
                    how = "fresh child processes after the failed hipGraph captures (whole step and segmented): eager issue, host bound"
                rc = 3
                if rank == 0:
                    if fb and "error" not in fb:
                        fb.setdefault("config", {})["capture_error"] = capture_error
                        fb["config"]["measured_in"] = how
                        fb["config"].pop("leg", None)
                        _flush_c_stdio()
                        print(json.dumps(fb), flush=True)
                        rc = 0
                    else:
                        sys.stderr.write("bench: the eager fall-back run failed too (%s)\n" % (fb or {}).get("error"))
                else:
                    rc = 0
                sys.exit(rc)
        if capture_error is None:
            def step():                  # noqa: F811
                loss = replay()
                host_step_end()
                return loss
        else:
            sys.stderr.write("bench: hipGraph capture failed (%s); falling back to eager issue on every rank\n" % capture_error)
            step, use_graph = eager_step, False
        for _ in range(args.warmup):
            loss = step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        loss = step()
    e1.record()
    barrier()
    dt = time.perf_counter() - t0
    final_loss = float(loss.item())
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_step = dt * 1000.0 / args.steps
    rows_s = B * world * args.steps / dt
    capture_error_msg = capture_error
    rccl_info = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:          # noqa: BLE001
            ver = None
        seen = torch.ones(1, device=device)
        dist.all_reduce(seen)                                  # = number of ranks the communicator really spans
        rccl_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen_by_allreduce": int(seen.item()),
                     "rccl_version": ver, "payload_dtype": compress or "fp32", "payload_summed_in": compress or "fp32",
                     "slices_per_step": len(pipe.slices) if pipe is not None else None,
                     "one_gpu_validation_mode": bool(one_gpu)}

    # host issue time of one step (no sync inside): if this approaches ms_per_step the run is launch bound
    torch.cuda.synchronize()
    th = time.perf_counter()
    eager_step()
    host_ms = (time.perf_counter() - th) * 1e3
    torch.cuda.synchronize()

    roofline, breakdown = None, None
    if rank == 0 and not args.no_breakdown:
        # The instrumented step is issued eagerly (events cannot be read back from a replayed graph).  Eager issue is host
        # bound, so each event pair would also bracket the launch latency of an idle stream; a spin kernel on the main
        # stream (the other streams wait on it) holds the GPU back until the host has queued the whole step, and the pairs
        # then bracket GPU execution only -- the per-kernel averages agree with rocprofv3's for the same command.
        def spin_ms(cycles):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); torch.cuda._sleep(cycles); b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b)
        spin_ms(1000)
        per_ms = 2_000_000 / max(spin_ms(2_000_000), 1e-3)
        # ... and it runs SERIALIZED (one stream, no backward pipeline): with three streams in flight an event pair would
        # also count the time a kernel waits for CUs held by the other streams' kernels, which is not its duration.
        # The single-GPU pipeline stays attached, on the main stream: its slice's weight-gradient launch is the one that also applies
        # AdamW (gstvd_gemm_grouped_adamw) -- the instrumented pass must time the kernels the timed step launches.
        eng = model.engine
        keep_pipe = eng.pipe is not None and not eng.pipe.collective
        saved = (eng.use_streams, eng.pipe, getattr(eng, "aux", None), eng.pipe.update_stream if eng.pipe is not None else None)
        eng.use_streams = False
        if keep_pipe:
            eng.aux, eng.pipe.update_stream = torch.cuda.current_stream(), False
        else:
            eng.pipe = None
        try:
            eager_step()                                   # re-plans the arena / tables for the serialized schedule
            with ops.Profiler() as prof:
                torch.cuda._sleep(int(per_ms * (host_ms + 10.0)))
                eager_step()
            agg = prof.summary()
            co = prof.summary(scope="coattn")
        finally:
            eng.use_streams, eng.pipe = saved[0], saved[1]
            if saved[2] is not None:
                eng.aux = saved[2]
            if eng.pipe is not None:
                eng.pipe.update_stream = saved[3]
        gemms = {k[5:]: v for k, v in agg.items() if k.startswith("gemm:")}      # keyed by the launched kernel's mangled symbol
        # Dominant kernel = the GEMM instantiation with the largest total duration in this step, measured live with HIP
        # events on the stream each launch goes to (the committed rocprofv3 stats of the same command rank the same
        # kernels on top: profiles/r01_kernel_stats_bench.csv).  The next two are listed beside it.
        order = sorted(gemms, key=lambda k: -gemms[k]["ms"])
        dom = order[0]
        dv = gemms[dom]
        ach = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
        all_gemm_flops = sum(v["flops"] for v in gemms.values())
        all_gemm_ms = sum(v["ms"] for v in gemms.values())
        pmc_key = lambda sym: ("gemm_grouped_adamw_256" if "grouped_adamw" in sym else "gemm_grouped_wgrad_256" if "grouped" in sym else "gemm_pc256" if "pc256" in sym else "gemm_dma256" if "gemm_dma256" in sym
                               else "gemm_dma128" if "Li128ELi128E" in sym else "gemm_dma64")
        traffic, traffic_src = None, None
        for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:   # HBM bytes per launch from the committed PMC passes (tools/pmc_bench.sh; FETCH_SIZE x2 gfx950 correction +
                   # WRITE_SIZE): an average over the launches of the same tile CLASS, read from a file -- not measured in this run
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    traffic = json.load(f).get(pmc_key(dom), {}).get("hbm_bytes_per_launch")
                traffic_src = "profiles/%s (tile-class average of a separate rocprofv3 --pmc run)" % fn
                break
            except (OSError, ValueError):
                continue
        # the committed rocprofv3 --kernel-trace --stats average of the SAME kernel inside replayed steps (profiles/rNN_kernel_stats_bench.csv):
        # the serialized eager pass above reads 2-7 % slower than that (profiles/r05_bench_repeat.txt), so the two are printed side by side
        rocprof_us, rocprof_src = None, None
        for fn in ("r06_kernel_stats_bench.csv", "r05_kernel_stats_bench.csv"):
            try:
                import csv as _csv
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    for r in _csv.DictReader(f):
                        nm = r.get("Name", "")
                        if nm == dom or nm == (demangle(dom) or "") or ("grouped_adamw" in dom and "grouped_adamw" in nm):
                            rocprof_us = round(float(r["AverageNs"]) / 1e3, 2)
                            break
                if rocprof_us is not None:
                    rocprof_src = "profiles/%s (rocprofv3 --kernel-trace --stats of bench.py, committed; not measured in this run)" % fn
                    break
            except (OSError, ValueError, KeyError):
                continue
        others = [{"kernel": k, "achieved": round(gemms[k]["flops"] / (gemms[k]["ms"] * 1e-3) / 1e12, 2),
                   "frac": round(gemms[k]["flops"] / (gemms[k]["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                   "ms_per_step": round(gemms[k]["ms"], 3)} for k in order[1:3]]
        # which roofline bounds the dominant kernel: the larger of its algorithmic times.  The weight-gradient launch that also
        # applies AdamW moves 26 B per weight + its operands (12 GB per step) for 1.4 PFLOP: HBM, not MFMA, is its floor.
        t_mfma = dv["flops"] / (PEAK_BF16_TFLOPS * 1e12)
        t_hbm = (dv["bytes"] or 0.0) / (PEAK_HBM_GBS * 1e9)
        if t_hbm > t_mfma:
            gbs = dv["bytes"] / (dv["ms"] * 1e-3) / 1e9
            head = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "mfma_achieved_tflops": round(ach, 2), "mfma_frac": round(ach / PEAK_BF16_TFLOPS, 4)}
        else:
            head = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4)}
        roofline = dict(head)
        roofline.update({"traffic": traffic, "traffic_source": traffic_src,
                    "kernel": dom, "kernel_demangled": demangle(dom), "launches_per_step": dv["launches"],
                    "flops_per_launch": dv["flops"] / dv["launches"], "avg_launch_us": round(1e3 * dv["ms"] / dv["launches"], 2),
                    "avg_launch_us_rocprof": rocprof_us, "avg_launch_us_rocprof_source": rocprof_src,
                    "frac_at_rocprof_duration": (round(dv["bytes"] / dv["launches"] / (rocprof_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)
                                                 if (rocprof_us and dv["bytes"] and t_hbm > t_mfma) else None),
                    "algorithmic_bytes_per_launch": (dv["bytes"] / dv["launches"]) if dv["bytes"] else None,
                    "all_gemm_tflops": round(all_gemm_flops / (all_gemm_ms * 1e-3) / 1e12, 2),
                    "all_gemm_ms_per_step": round(all_gemm_ms, 3),
                    "step_algorithmic_tflops": round(rows_s / world * FLOP_PER_ROW_TRAIN / 1e12, 2),
                    "step_frac": round(rows_s / world * FLOP_PER_ROW_TRAIN / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "next_kernels": others})
        # north_star's own target: MFMA utilisation of the co-attention block (BertConnectionLayer x6, vilbert_dialog.py:646-773):
        # every GEMM launched inside Engine.conn_layer and its backward (QKV1/QKV2, biOutput dense1/2, both FFNs, their input
        # gradients; the weight gradients run in the grouped launch and are listed under next_kernels) -- FLOPs / event time / peak
        cg = {k: v for k, v in co.items() if k.startswith("gemm:")}
        ca = {k: v for k, v in co.items() if k.startswith("attn_")}
        if cg:
            cg_fl, cg_ms = sum(v["flops"] for v in cg.values()), sum(v["ms"] for v in cg.values())
            ca_fl, ca_ms = sum(v["flops"] for v in ca.values()), sum(v["ms"] for v in ca.values())
            roofline["coattn_frac"] = round(cg_fl / (cg_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
            roofline["coattn"] = {"gemm_tflops": round(cg_fl / (cg_ms * 1e-3) / 1e12, 1), "gemm_ms_per_step": round(cg_ms, 3),
                                  "gemm_launches": sum(v["launches"] for v in cg.values()),
                                  "attention_tflops": round(ca_fl / (ca_ms * 1e-3) / 1e12, 1) if ca_ms else None,
                                  "attention_ms_per_step": round(ca_ms, 3),
                                  "block_frac_incl_attention": round((cg_fl + ca_fl) / ((cg_ms + ca_ms) * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "note": "forward + input-gradient GEMMs of the 6 connection layers; target >= 0.40 (north_star)"}
        # "HBM GB/s of the bandwidth-bound kernels vs peak" -- algorithmic bytes per launch (the byte counts
        # ops.py states next to each call: AdamW 30 B/param, LayerNorm 3 / 5 activations per element, CE logits once / twice)
        # over the HIP-event duration of the same serialized pass, against the 8 TB/s HBM3E peak
        hbm = []
        for k in ("adamw", "ln_fwd", "ln_bwd", "ce_fwd", "ce_bwd"):
            v = agg.get(k)
            if v and v["bytes"] and v["ms"] > 0:
                gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                hbm.append({"kernel": k, "launches": v["launches"], "ms_per_step": round(v["ms"], 3),
                            "algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"]), "achieved": round(gbs, 1), "unit": "GB/s",
                            "peak": PEAK_HBM_GBS, "frac": round(gbs / PEAK_HBM_GBS, 4)})
        roofline["hbm"] = hbm
        roofline["launching_calls_per_step"] = sum(v["launches"] for v in agg.values())
        breakdown = {k: dict(launches=v["launches"], ms=round(v["ms"], 3),
                             tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                             gbps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None)
                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        if args.breakdown_json:
            shapes = prof.summary(by_shape=True)
            shp = {k: dict(launches=v["launches"], ms=round(v["ms"], 3), us_per_launch=round(1e3 * v["ms"] / v["launches"], 1),
                           tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None)
                   for k, v in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])}
            with open(args.breakdown_json, "w") as f:
                json.dump({"ms_per_step": ms_step, "kernels": breakdown, "shapes": shp}, f, indent=1)

    # ---- PCIe-inclusive rate (reported beside `value`, never as `value`): every step's rows come from pageable host memory
    # through step.PinnedStager (pinned slots, copy stream, one batch ahead) and are refreshed IN PLACE into the tensors the
    # captured graph reads
    pcie = None
    if rank == 0 and world == 1 and not args.no_h2d:
        from gst_visdial_amd.step import PinnedStager
        stager = PinnedStager(device, depth=2)
        host_rows = [{k: v.cpu() for k, v in synthetic_rows(B, T, R, U, F, V, 4000 + i, "cpu").items()} for i in range(3)]
        nb = sum(v.numel() * v.element_size() for v in host_rows[0].values())

        def h2d_steps(n):
            pending = stager.fill(host_rows[0])
            for i in range(n):
                rows = stager.upload(pending)
                for k, v in rows.items():
                    batch[k].copy_(v)
                step()
                pending = stager.fill(host_rows[(i + 1) % 3])          # host half of the next batch under this step's replay
            stager.upload(pending)
        h2d_steps(2)
        torch.cuda.synchronize()
        th = time.perf_counter()
        n_h2d = max(5, min(args.steps, 20))
        h2d_steps(n_h2d)
        torch.cuda.synchronize()
        dth = time.perf_counter() - th
        pcie = {"value": round(B * n_h2d / dth, 3), "unit": "dialog-rounds/sec", "ms_per_step": round(dth * 1e3 / n_h2d, 3),
                "host_bytes_per_step": nb, "how": "pageable host rows -> pinned slots, single-threaded host fill + H2D on a copy stream one batch ahead (under the "
                "replay) -> in-place refresh of the captured graph's inputs (gst_visdial_amd.step.PinnedStager, mode %s)" % stager.mode}

    # ---- the parity mode's cost: the same step with every GEMM on the exact-fp32 MFMA (what the 1e-4 logit tolerance is
    # tested in); eager issue, a few steps
    fp32_ms = None
    if rank == 0 and world == 1 and not args.no_fp32 and args.precision == "bf16":
        try:
            m32, _ = build_model(device, "fp32", seed=1234, streams=not args.no_streams)
            m32.train()
            o32 = FusedAdamW(m32, lr=2e-5, warmup_steps=1500, t_total=100000)

            def s32():
                l32, _ = m32(**batch)
                l32.backward()
                o32.step()
                o32.zero_grad()
            s32(); s32()
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            for _ in range(3):
                s32()
            torch.cuda.synchronize()
            fp32_ms = round((time.perf_counter() - t32) * 1e3 / 3, 2)
            del m32, o32
            torch.cuda.empty_cache()
        except Exception as ex:            # noqa: BLE001 -- a diagnostic beside the headline must not take the line down
            sys.stderr.write("bench: fp32 parity-mode timing skipped (%s: %s)\n" % (type(ex).__name__, ex))

    # ---- BASELINE configs[3] beside the headline: sampling decode + candidate scoring at full size (outside the timed region)
    side = None
    if rank == 0 and world == 1 and not args.no_eval_decode and args.precision == "bf16":
        try:
            side = eval_decode_side(device, V)
        except Exception as ex:            # noqa: BLE001 -- a side measurement must not take the line down
            sys.stderr.write("bench: configs[3] side measurements skipped (%s: %s)\n" % (type(ex).__name__, ex))
            side = {"error": "%s: %s" % (type(ex).__name__, ex)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, T, R, U, F, V)

    # ---- rows sensitivity: the same step at 32 and 64 rows per GPU, each in a fresh child process; a SIDE
    # record (never `value`): step_frac / coattn_frac / all_gemm_tflops per row count
    rows_sens = None
    if (rank == 0 and world == 1 and not force_dist and args.leg is None and B == 16 and args.precision == "bf16" and roofline is not None
            and not args.no_rows_sensitivity):
        rows_sens = {"16": {"ms_per_step": round(ms_step, 3), "rounds_per_s": round(rows_s, 1), "step_frac": roofline.get("step_frac"),
                            "coattn_frac": roofline.get("coattn_frac"), "all_gemm_tflops": roofline.get("all_gemm_tflops")}}
        import subprocess              # (288 GB of HBM: the children fit beside this process's model)
        for rr in (32, 64):
            cmd = [sys.executable, os.path.abspath(__file__), "--rows-per-gpu", str(rr), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                   "--no-eval-decode", "--no-fp32", "--no-h2d", "--no-rows-sensitivity", "--leg", "rows_sens%d" % rr]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
                line, _ = parse_last_json(r.stdout.decode("utf-8", "replace"))
                rf = (line or {}).get("roofline") or {}
                rows_sens[str(rr)] = ({"ms_per_step": line.get("ms_per_step"), "rounds_per_s": round(line.get("value"), 1), "step_frac": rf.get("step_frac"),
                                       "coattn_frac": rf.get("coattn_frac"), "all_gemm_tflops": rf.get("all_gemm_tflops")}
                                      if line and r.returncode == 0 else {"error": "child exit code %d" % r.returncode})
            except Exception as ex:        # noqa: BLE001 -- a side record must not take the line down
                rows_sens[str(rr)] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        rows_sens["reading"] = ("coattn_frac = bf16 MFMA utilisation of the 6 connection layers' forward + input-gradient GEMMs (north_star: >= 0.40); "
                                "never `value`: the headline stays BASELINE configs[1]'s 16 rows per GPU")

    if rank == 0:
        out = {"metric": "dialog-rounds/sec (enc_dec_a train step)", "value": round(rows_s, 3), "unit": "dialog-rounds/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
               "data": "synthetic (random-init weights, synthetic 10-round-dialog rows, features resident in HBM)",
               "config": {"workload": "enc_dec_a train step (fwd+loss+bwd+allreduce+AdamW, dropout on), %d rows/GPU, "
                                      "seq_len %d, 37x2048 region features, answer len 25 [%s]"
                                      % (B, T, "BASELINE configs[1] per-GPU shape" if B == 16 else
                                         "BASELINE configs[2] per-rank shape (global 80 at 8 GPUs)" if B == 10 else "custom rows/GPU"),
                          "global_batch": B * world, "rows_per_gpu": B, "seq_len": T, "parallelism": "dp%d" % world, "leg": args.leg,
                          "gpu_ms_per_step_events": round(e0.elapsed_time(e1) / args.steps, 3), "eager_host_issue_ms_per_step": round(host_ms, 3), "hip_graph": (use_graph if use_graph == "segmented" else bool(use_graph)), "graphs_per_step": (getattr(replay, "n_graphs", 1) if use_graph else 0), "capture_error": capture_error_msg,
                          "final_loss": round(final_loss, 4)},
               "roofline": roofline, "cpu_baseline": cpu}
        if breakdown is not None:
            out["kernel_breakdown_ms"] = {k: v["ms"] for k, v in list(breakdown.items())[:12]}
        collective = world > 1 or force_dist
        out["config"]["grad_allreduce_dtype"] = (compress or "fp32") if collective else None
        # how the optimizer step is applied: inside the weight-gradient launch (single GPU: gstvd_gemm_grouped_adamw + remainder pass,
        # bit-identical to the two launches) or as a pass of its own behind the slice's all-reduce (N>1)
        out["config"]["optimizer_update"] = (None if pipe is None else
                                             "in the weight-gradient launch's epilogue + remainder pass" if pipe.fuse_handle() is not None
                                             else "sharded: reduce-scatter -> AdamW on the rank's 1/N shard -> all-gather of the bf16 shadow weights"
                                             if (pipe.shard_update and pipe.collective) else "AdamW pass per gradient slice (full, on every rank)")
        out["config"]["gradient_slices_per_step"] = len(pipe.slices) if pipe is not None else None
        # N > 1, bf16 payload: the weight-gradient launch writes the payload of the GEMM weights itself; only the rest of a slice is cast
        out["config"]["payload_written_by_wgrad_launch"] = bool(pipe is not None and pipe.Gb is not None and any(len(k[2]) for k in pipe._cast_plans))
        # what the communicator really is (N>1 only runs on the driver's node: this is the evidence that it was RCCL, over how
        # many ranks, with which payload); summing bf16 payloads IN bf16 deviates from the reference's fp32 reduce-add by
        # <= 4e-3 of a tensor's norm at 8 ranks (tests/test_dp_gloo.py::test_eight_rank_graded_slices_bf16_payload_error_bound)
        out["config"]["rccl"] = rccl_info if collective else None
        out["config"]["dropout_seed_per_rank"] = bool(params.get("amd_seed_per_rank", True)) if collective else None
        # what the driver needs to turn per-N values into a scaling curve (weak scaling: rows/GPU fixed, so scaling(N) = value_N / value_1)
        out["config"]["scaling_inputs"] = {"rows_per_gpu": B, "global_batch": B * world, "payload_dtype": (compress or "fp32") if collective else None,
                                           "scaling_vs_n1": "value(N) / value(1) at equal rows_per_gpu; bench.py never reports efficiency itself"}
        from gst_visdial_amd import graph as _g
        out["config"]["capture_quiesce"] = _g.LAST_QUIESCE[0] if use_graph else None
        out["config"]["rows_sensitivity"] = rows_sens
        out["config"]["fp32_parity_mode_ms_per_step"] = fp32_ms
        out["config"]["pcie_inclusive"] = pcie
        out["config"]["decode"] = (side or {}).get("decode") if side and "error" not in side else side
        out["config"]["score"] = (side or {}).get("score") if side and "error" not in side else None
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()          # RCCL prints its banner here; the JSON line must come last
    # ---- extra legs beside the headline (N > 1): the same step at 10 rows/rank (BASELINE configs[2]: global 80 at 8 GPUs) and with
    # the reference-faithful fp32 gradient all-reduce (train_gen.py:324 reduce-adds fp32; the headline's payload is bf16).  Fresh
    # child processes per rank, after this process's group is gone; a failed leg is reported as {"error": ...} in its slot.
    if args.leg is None and (args.legs == "on" or (args.legs == "auto" and world > 1)):
        legs = []
        if B != 10:
            legs.append(("rows10", ["--rows-per-gpu", "10"]))
        if compress is not None:
            legs.append(("fp32_allreduce", ["--grad-compress", "none"]))
        if args.shard_update != "on":
            # the same step with the optimizer sharded over the ranks (1/N of the AdamW pass per rank, same bytes on the links)
            legs.append(("sharded_update", ["--shard-update", "on"]))
        for k, (name, extra) in enumerate(legs):
            res = run_leg(name, extra, argv, rank, world, local, leg_port(1 + k))
            if rank == 0:
                if res and "error" not in res:
                    c = res.get("config", {})
                    res = {"value": res.get("value"), "unit": res.get("unit"), "ms_per_step": res.get("ms_per_step"), "n_gpus": res.get("n_gpus"),
                           "rows_per_gpu": c.get("rows_per_gpu"), "global_batch": c.get("global_batch"), "hip_graph": c.get("hip_graph"),
                           "grad_allreduce_dtype": c.get("grad_allreduce_dtype"), "optimizer_update": c.get("optimizer_update"),
                           "ranks_seen_by_allreduce": (c.get("rccl") or {}).get("ranks_seen_by_allreduce"), "workload": c.get("workload")}
                out["config"].setdefault("legs", {})[name] = res
                if name == "rows10" and world == 8:
                    out["config"]["global80"] = res      # BASELINE configs[2] exactly: batch 80 global over 8 GPUs
    _flush_c_stdio()                          # ... and it sits in the C stdio buffer: push it out before our line
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    # RCCL's version banner is written by a helper thread and can arrive late (seen once in ~10 runs: after the JSON, at
    # exit-time flush).  The contract is "the JSON line is the last line": whatever C code writes to stdout from here on
    # goes nowhere.
    _flush_c_stdio()
    try:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    except OSError:
        pass


if __name__ == "__main__":
    main()
