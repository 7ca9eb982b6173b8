#!/usr/bin/env python
"""bench.py -- forward-scoring throughput (nucleotides/s) of the MI355X-native StripedHyena engine.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
prints ONE JSON line on rank 0.

Workloads (BASELINE.json):
  primary  `value`    : configs[1]  evo-1-8k-base scoring, batch 8 x 8,192 nt (T = 8,193 with BOS), bf16, per GPU.
                        N > 1: every rank scores its own batch of 8 (independent sequences, no data-path
                        collective) -> weak scaling.
  secondary `ctx131k` : configs[2]/[3]  evo-1-131k-base scoring at 131,072 nt: batch 1 on one GPU; for N > 1
                        batch N with the SEQUENCE sharded over the N ranks (RCCL all-gather of Hyena end states
                        and of K/V, 2-row halo exchange) -- reported inside the same line.
A step = one scoring pass (forward + log-prob reduction) over one resident batch of synthetic ACGT;
tokenisation, weight creation and H2D copies are outside the timed region.  Weights are synthetic (real
shapes, random init -- no checkpoint is reachable offline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def acgt_ids(batch, nt, seed0, device):
    rows = []
    for b in range(batch):
        rng = np.random.default_rng(seed0 + b)
        rows.append(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=nt))
    ids = np.concatenate([np.zeros((batch, 1), np.int64), np.stack(rows).astype(np.int64)], axis=1)   # BOS
    return torch.from_numpy(ids).to(device)


def build_model(name, device, seed=0):
    from evo_amd.models import _CONFIG_FOR, load_config
    from evo_amd.sh.model import StripedHyena
    from evo_amd.synthetic import synthetic_state_dict
    m = StripedHyena(load_config(_CONFIG_FOR[name]))
    m.load_state_dict(synthetic_state_dict(m, seed=seed, device=device), strict=True)
    m.to_bfloat16_except_poles_residues()
    return m.to(device)


def scoring_step(model, ids):
    """forward + per-token log-prob of the next token: exactly what evo_amd.score_sequences runs on the device
    (32 blocks, final norm, fused unembed + log-softmax + gather)."""
    from evo_amd.scoring import score_logprobs_device
    return score_logprobs_device(model, ids)[0]


def timed(fn, steps, warmup, dist_on, stats=None):
    """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; wall clock, MAX
    over ranks (the contract).  Beside it, a HIP event is recorded on the launch stream at every step boundary:
    `stats` receives the per-step device durations (median / min / max) -- SURVEY 8(d) asks for hipEvent medians."""
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        fn()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if stats is not None:
        per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        stats.update({"hip_event_ms_median": per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2]),
                      "hip_event_ms_min": per[0], "hip_event_ms_max": per[-1],
                      "wall_ms_mean": dt / steps * 1e3})
    return dt


def pmc_traffic(kernel, B, T, with_source=False):
    """HBM bytes per launch from rocprofv3 PMC passes (read requests x 128 B + 64-byte writes x 64 + 32-byte writes x 32: MI355X_MICROARCH.md).
    PMC counters cannot be collected from inside this process.  tools/gpu_check.sh runs the two counter passes right BEFORE this script and
    leaves gpurun_out/check/pmc_traffic_live.json (same box, same tree: "this run"); without it the committed record
    profiles/pmc_traffic.json is used.  `with_source` also returns (commit the passes were taken at, "live" | "recorded")."""
    for path, kind in ((os.path.join(ROOT, "gpurun_out", "check", "pmc_traffic_live.json"), "live"),
                       (os.path.join(ROOT, "profiles", "pmc_traffic.json"), "recorded")):
        try:
            with open(path) as f:
                d = json.load(f)
            v = d.get(kernel, {}).get(f"B{B}_T{T}")
            if v is not None:
                return (v, d.get("_commit", {}).get(kernel), kind) if with_source else v
        except (OSError, ValueError):
            continue
    return (None, None, None) if with_source else None


def hyena_roofline(ops, model, ksum, B, T, alg_bytes, device):
    """`roofline` of the Hyena operator (the north-star's HBM-bound kernel).  Default engine: ONE launch per layer
    (hyena_ct_kernel: z read once, y written once = the algorithmic bytes; scoring, cached prefill and the sequence-parallel
    shards alike).  The three-launch modal form -- the round-1 operator, still used for padding masks and very short inputs -- is
    timed beside it on the same shape with one layer's filter, so that both fractions are live numbers of this run."""
    from evo_amd.ops import KernelTimer
    io_live = dict(getattr(ops, "last_hyena_io", {}))       # of the timed steps (the reference run below overwrites it)
    blk = model.blocks[model.hyena_layer_idxs[0]]
    f = blk.filter
    z = torch.randn(B, T, 3 * 4096, device=device).to(torch.bfloat16)
    keep, ops.timer = ops.timer, KernelTimer()
    try:
        for _ in range(3):
            ops.hyena_prefill(z, f._fir_w, f.short_filter_bias, f._poles, f._residues, f.D, model.num_heads)
        torch.cuda.synchronize()
        modal = ops.timer.summary()
    finally:
        ops.timer = keep
    apply_ms = modal["hyena_apply"][1]
    op3_ms = apply_ms + modal["hyena_seg_state"][1] + modal["hyena_carry_scan"][1]
    three = {"hyena_apply_ms": apply_ms, "apply_frac": alg_bytes / (apply_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "operator_3_launch_ms": op3_ms, "operator_frac": alg_bytes / (op3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "apply_traffic": pmc_traffic("hyena_apply_kernel", B, T)}
    if "hyena_mfma" in ksum:
        ms = ksum["hyena_mfma"][1]
        # the bytes of the tokens THIS launch walks: with ops.hyena_tail_split the last token of every row (T = 512 k + 1) is not in it
        # (it takes the fused single-token launch, `tail_split` below), so B * 512 k tokens x 32,768 B, not B * T
        alg_bytes = io_live.get("mfma") or alg_bytes
        ach = alg_bytes / (ms * 1e-3) / 1e9
        tail = None
        if "gemv_hyena" in ksum and getattr(ops, "hyena_tail_split", False):
            tail = {"what": "the token behind the whole tiles of every batch row: pre-norm + projections + FIR / modal step in the fused single-token "
                            "launch of the decode path, from the operator's end state (instead of a ragged tile with one valid step)",
                    "launches_per_layer": 1, "avg_ms": ksum["gemv_hyena"][1]}
        traffic, t_commit, t_kind = pmc_traffic("hyena_ct_kernel", B, T, with_source=True)
        return {"kernel": "hyena_ct_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": (f"rocprofv3 PMC passes of THIS gpu_check run, right before this script (commit {t_commit})" if t_kind == "live" else
                                   f"recorded rocprofv3 PMC passes of commit {t_commit} (profiles/pmc_traffic.json), not this run"),
                "z_layout": "channel-major z^T [3 D][B Tp], written by the projection's dense layer launched with swapped operands (a lane's eight "
                            "steps of a channel = 16 contiguous bytes, loaded straight into registers: no window in LDS)",
                "y_layout": "blocked [B T / 128][D / 16][128][16] (whole cache lines per store; the output projection's dense layer gathers it)",
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms,
                "tensor_bytes_per_launch": io_live.get("mfma"),
                "operator_frac": ach / HBM_PEAK_GBS, "tail_split": tail, "modal_three_launch": three,
                "note": "one launch IS the whole operator (z read once, y written once: 32,768 B per token and layer); round 1 reported "
                        "hyena_apply_kernel, one of three launches: compare frac with modal_three_launch.operator_frac of the same run"}
    achieved = alg_bytes / (ksum["hyena_apply"][1] * 1e-3) / 1e9
    op_ms = ksum["hyena_apply"][1] + ksum["hyena_seg_state"][1] + ksum["hyena_carry_scan"][1]
    return {"kernel": "hyena_apply_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("hyena_apply_kernel", B, T),
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ksum["hyena_apply"][1],
            "tensor_bytes_per_launch": io_live.get("apply"),
            "operator_3_launch_ms": op_ms, "operator_frac": alg_bytes / (op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def flops_per_token(T):
    """GEMM 12.889 GF + unembed 4.19 MF + causal attention 24,576*T."""
    return 32 * 402_784_256 + 2 * 4096 * 512 + 24_576 * T


# ---------------------------------------------------------------------------------------------------------------------------------------
# `box`: what THIS box gives.  The driver's headline fell 5.6 % between rounds 4 and 5 on a box where an unchanged
# kernel ran 9 % slower: a bench line needs the box's own rates beside it.  Measured in this process right before the headline:
#   * hbm_copy_GBs        evo_probe_copy_f4 over 1 GiB (read + write counted: 2 GiB per pass), HIP events
#   * mfma_probe_tflops   evo_probe_mfma_bf16: register-resident 16x16x32 bf16 MFMA stream, 256 x 4 waves, ~0.2 s (long enough for the
#                         power management to settle the clock), pseudo-random operands
#   * library_gemm_tflops torch.mm (hipBLASLt, fixed by the image) 8,192^3 bf16 on N(0, 1) operands: the dense-layer rate this box sustains
#                         with code that is not ours
#   * clocks / power      /sys/class/drm/card*/device/hwmon (power average + cap, sclk, mclk), sampled by a thread over the timed region
# `value_per_calibrated_box` = value / (share_dense * library_gemm / REF_GEMM + (1 - share_dense) * hbm_copy / REF_COPY): the headline on a
# box that gives exactly the reference rates (REF_GEMM 1,500 TFLOP/s = the middle of the 1.42-1.6 PFLOP/s hipBLASLt range of rounds 2-5,
# REF_COPY 6,290 GB/s = MI355X_MICROARCH.md's float4 copy); share_dense = this run's dense-layer share of the step.
REF_GEMM_TFLOPS = 1500.0
REF_COPY_GBS = 6290.0


def _gpu_sysfs_dir(index=0):
    """The hwmon directory of HIP device `index`, matched by PCI address (a container sees every card of the host under /sys/class/drm
    but only its own GPUs through HIP), or None."""
    import glob
    cache = _gpu_sysfs_dir.__dict__.setdefault("cache", {})
    if index in cache:
        return cache[index]
    want = None
    try:
        pr = torch.cuda.get_device_properties(index)
        want = "%04x:%02x:%02x.0" % (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
    except Exception:  # noqa: BLE001
        pass
    hit = None
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.realpath(os.path.join(card, "device"))
        hw = sorted(glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")))
        if not hw:
            continue
        if want is not None and os.path.basename(dev).lower() == want.lower():
            hit = hw[0]
            break
    cache[index] = hit
    return hit


def _read_int(path):
    try:
        with open(path) as fh:
            return int(fh.read().strip())
    except (OSError, ValueError):
        return None


def gpu_telemetry(index=0):
    """One reading: {power_W, power_cap_W, sclk_MHz, mclk_MHz, temp_C}; missing entries are None (sysfs layouts differ)."""
    d = _gpu_sysfs_dir(index)
    if d is None:
        return {}
    pw = _read_int(os.path.join(d, "power1_average"))
    if pw is None:
        pw = _read_int(os.path.join(d, "power1_input"))
    cap = _read_int(os.path.join(d, "power1_cap"))
    f1, f2 = _read_int(os.path.join(d, "freq1_input")), _read_int(os.path.join(d, "freq2_input"))
    tp = _read_int(os.path.join(d, "temp1_input"))
    return {"power_W": None if pw is None else pw / 1e6, "power_cap_W": None if cap is None else cap / 1e6,
            "sclk_MHz": None if f1 is None else f1 / 1e6, "mclk_MHz": None if f2 is None else f2 / 1e6,
            "temp_C": None if tp is None else tp / 1e3}


class TelemetrySampler:
    """Samples gpu_telemetry() every `period` s on a host thread while a timed region runs (the reads are sysfs files: no GPU work)."""

    def __init__(self, index=0, period=0.05):
        import threading
        self.index, self.period, self.rows = index, period, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            r = gpu_telemetry(self.index)
            if r:
                self.rows.append(r)
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(timeout=2.0)

    def summary(self):
        out = {"samples": len(self.rows)}
        for k in ("power_W", "sclk_MHz", "mclk_MHz", "temp_C"):
            v = [r[k] for r in self.rows if r.get(k) is not None]
            if v:
                out[k + "_mean"] = sum(v) / len(v)
                out[k + "_min"], out[k + "_max"] = min(v), max(v)
        caps = [r["power_cap_W"] for r in self.rows if r.get("power_cap_W") is not None]
        if caps:
            out["power_cap_W"] = caps[0]
        return out


def box_probes(ops, device, local=0):
    """The box's own rates, measured now (see the comment above).  Every probe: 2 warm-up launches, then HIP events around the timed ones."""
    st = torch.cuda.current_stream().cuda_stream
    out = {}

    def ev_ms(fn, reps):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 256)
    dst = torch.empty_like(src)

    def copy():
        rc = ops.lib.evo_probe_copy_f4(src.data_ptr(), dst.data_ptr(), n, st)
        assert rc == 0, rc
    ms = ev_ms(copy, 20)
    out["hbm_copy_GBs"] = 2.0 * n / (ms * 1e-3) / 1e9
    out["hbm_copy_bytes"] = n
    del src, dst
    sink = torch.empty(256 * 256, dtype=torch.float32, device=device)
    iters = 2000000                                      # 256 x 4 waves x 2,000,000 x 16 MFMAs of 16,384 flop = 5.4e14 flop: ~0.3-0.4 s per launch

    def mf():
        rc = ops.lib.evo_probe_mfma_bf16(sink.data_ptr(), 256, iters, st)
        assert rc == 0, rc
    with TelemetrySampler(local) as tel:
        ms = ev_ms(mf, 1)
    out["mfma_probe_tflops"] = 256 * 4 * iters * 16 * 16384.0 / (ms * 1e-3) / 1e12
    out["mfma_probe_ms"] = ms
    out["mfma_probe_telemetry"] = tel.summary()
    g = torch.Generator(device=device).manual_seed(7)
    a_ = torch.randn(8192, 8192, generator=g, device=device).bfloat16()
    b_ = torch.randn(8192, 8192, generator=g, device=device).bfloat16()
    c_ = torch.empty(8192, 8192, dtype=torch.bfloat16, device=device)
    with TelemetrySampler(local) as tel:
        ms = ev_ms(lambda: torch.mm(a_, b_.t(), out=c_), 200)
    out["library_gemm_tflops"] = 2.0 * 8192 ** 3 / (ms * 1e-3) / 1e12
    out["library_gemm_shape"] = "8192 x 8192 x 8192 bf16 (torch.mm -> hipBLASLt), N(0, 1) operands, 200 launches"
    out["library_gemm_telemetry"] = tel.summary()
    out["idle_telemetry"] = gpu_telemetry(local)
    out["reference"] = {"library_gemm_tflops": REF_GEMM_TFLOPS, "hbm_copy_GBs": REF_COPY_GBS}
    return out


def _host_mem_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(model=None, nt=512):
    """The oracle (a 'port' of the reference forward: its eager-attention CPU path) on the host cores, BASELINE
    configs[0]: the FULL 32-block evo-1-8k-base forward on one 512-nt sequence, on the very weights the GPU run
    used (copied to the host).  fp32 mode (bf16-rounded weights up-cast to fp32, MKL sgemm): torch's CPU bf16 path
    was slower still on the GPU box's host, so this choice FAVOURS the CPU.  One untimed short pass warms the
    thread pools, then one timed full pass (~10 s).  Falls back to a 4-of-32-block slice (x8) only when the host
    has no room for the 39 GB of host weights."""
    from oracle import stripedhyena_ref as R
    # 32 threads: with all 256 hardware threads of the GPU box's host the same work took 30x longer (oversubscribed
    # small ops); `cores` reports what was actually used
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ids = acgt_ids(1, nt, 1234, "cpu")
    if model is not None and _host_mem_gb() > 60.0:
        cfg = R.RefConfig()                                  # evo-1-8k-base dims (the dataclass defaults)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        m = R.RefStripedHyena(cfg, sd, "fp32")
        del sd
        with torch.inference_mode():
            m(ids[:, :33])                                   # warm-up (thread pools, oneDNN primitives)
            t0 = time.perf_counter()
            logits = m(ids)[0]
            dt = time.perf_counter() - t0
        lsm = torch.log_softmax(logits.double()[0, :-1], -1)
        score = lsm.gather(-1, ids[0, 1:, None]).mean().item()
        out = {"value": nt / dt, "unit": "nt/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"oracle fp32 mode (bf16-rounded weights), the full 32-block evo-1-8k-base forward on 1 x {nt} nt "
                         f"(BASELINE configs[0]), one timed pass", "seconds": dt, "score": score}
        try:
            out["legs"] = cpu_baseline_legs(m, R, engine=model)
        except Exception as e:  # noqa: BLE001   (the extra legs never take the headline baseline down with them)
            out["legs"] = {"error": f"{type(e).__name__}: {e}"}
        return out
    cfg = R.RefConfig(num_layers=4, attn_layer_idxs=(2,))
    sd = R.make_synthetic_state_dict(cfg, 0)
    m = R.RefStripedHyena(cfg, sd, "fp32")
    with torch.inference_mode():
        m(ids)
        t0 = time.perf_counter()
        reps = 0
        while reps < 12 and time.perf_counter() - t0 < 20.0:
            m(ids)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
    full = dt * (32 / 4)
    return {"value": nt / full, "unit": "nt/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 mode (bf16-rounded weights), 4 of 32 blocks (3 Hyena + 1 attention) at D=4096, 1 x {nt} nt, "
                      f"{reps} reps, time x8 for full depth (host memory too small for the full-depth copy)",
            "seconds_per_4_blocks": dt}


def cpu_baseline_legs(m, R, engine=None):
    """BASELINE.md section 2, configs[1], [2], [4] on the host cores -- bounded samples of the same oracle on the same weights
    (the full passes are ~107 TFLOP and ~2.1 PFLOP of fp32 on a host), every extrapolation labelled:
      configs[1]  B = 1, T = 8,193: blocks 0-2 (Hyena) and 8 (attention) at full width, timed; the pass = 29 x mean(Hyena
                  block) + 3 x attention block (the blocks of a kind cost the same); x 8 rows stated, not run
      configs[2]  B = 1, T = 131,073: ONE Hyena block at the full length, timed; one attention block = its dense layers at the
                  full length (timed) + eager softmax attention of 256 query rows (the LAST rows: every key visible) against all
                  131,073 keys, timed and scaled to the causal triangle (x T/2 / 256); pass = 29 x Hyena + 3 x attention
      configs[4]  128-token prefill, then 32 recurrent decode steps (oracle caches), tok/s
    """
    import numpy as np
    legs = {}
    cfg = m.cfg
    emb = m.w["embedding_layer.weight"]

    def stream(T, seed):
        rng = np.random.default_rng(seed)
        ids = torch.from_numpy(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=T - 1).astype(np.int64))
        return emb[torch.cat([torch.zeros(1, dtype=torch.long), ids])][None]          # [1, T, D]

    with torch.inference_mode():
        # ---- configs[1]
        T = 8193
        x = stream(T, 1234)
        t_h = []
        for i in (0, 1, 2):
            t0 = time.perf_counter()
            x = m.hyena_block(x, i, None)
            t_h.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        m.attn_block(x, 8, None)
        t_a = time.perf_counter() - t0
        full = 29 * (sum(t_h[1:]) / 2) + 3 * t_a                                  # (block 0 also warms the FFT plans: not counted)
        legs["configs1"] = {"value": (T - 1) / full, "unit": "nt/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": "B = 1 x 8,192 nt: Hyena blocks 1, 2 and attention block 8 of the 32 timed at full width, "
                                      "pass = 29 x Hyena + 3 x attention (extrapolated in depth); the 8 rows of configs[1] cost 8 x this",
                            "hyena_block_s": sum(t_h[1:]) / 2, "attn_block_s": t_a, "pass_s_extrapolated": full}
        # ---- configs[2]
        T = 131073
        x = stream(T, 1234)
        t0 = time.perf_counter()
        x1 = m.hyena_block(x, 0, None)
        t_h131 = time.perf_counter() - t0
        # attention block: dense layers over the full length (qkv projection, out projection, MLP) + sampled softmax attention
        pre = "blocks.8."
        t0 = time.perf_counter()
        qkv = m.linear(m.rmsnorm(x1, m.w[pre + "pre_norm.scale"]), m.w[pre + "inner_mha_cls.Wqkv.weight"],
                       m.w[pre + "inner_mha_cls.Wqkv.bias"]).reshape(1, T, 3, cfg.num_attention_heads, cfg.head_dim)
        u2 = m.linear(qkv[:, :, 0].reshape(1, T, -1), m.w[pre + "inner_mha_cls.out_proj.weight"],
                      m.w[pre + "inner_mha_cls.out_proj.bias"]) + x1
        m.mlp(m.rmsnorm(u2, m.w[pre + "post_norm.scale"]), pre)
        t_dense = time.perf_counter() - t0
        nq = 256
        t0 = time.perf_counter()
        m.attention(qkv[:, T - nq:, 0], qkv[:, :, 1], qkv[:, :, 2], q_pos0=T - nq)
        t_rows = time.perf_counter() - t0
        t_attn = t_rows * (T / 2.0) / nq                                           # causal triangle: mean T/2 keys per row
        full = 29 * t_h131 + 3 * (t_dense + t_attn)
        legs["configs2"] = {"value": (T - 1) / full, "unit": "nt/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": "EXTRAPOLATED: B = 1 x 131,072 nt: one Hyena block at the full length timed; one attention block = "
                                      "its dense layers at the full length (timed) + eager softmax attention of the last 256 query rows "
                                      "against all keys (timed), scaled to the causal triangle; pass = 29 x Hyena + 3 x attention",
                            "hyena_block_s": t_h131, "attn_dense_s": t_dense, "attn_256_rows_s": t_rows,
                            "attn_softmax_s_extrapolated": t_attn, "pass_s_extrapolated": full}
        del x, x1, qkv, u2
        # ---- configs[4]
        rng = np.random.default_rng(1234)
        ids = torch.from_numpy(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=128).astype(np.int64))[None]
        ipd = m.initialize_inference_params()
        logits, ipd = m(ids, ipd)
        ipd["mha"].seqlen_offset = ids.shape[1]
        ipd["hyena"].seqlen_offset = ids.shape[1]
        tok = logits[:, -1].argmax(-1, keepdim=True)
        n_dec = 32
        o_toks, o_logits = [tok], [logits[0, -1].float()]
        t0 = time.perf_counter()
        for _ in range(n_dec):
            logits, ipd = m(tok, ipd)
            ipd["mha"].seqlen_offset += 1
            ipd["hyena"].seqlen_offset += 1
            tok = logits[:, -1].argmax(-1, keepdim=True)
            o_toks.append(tok)
            o_logits.append(logits[0, -1].float())
        dt = time.perf_counter() - t0
        legs["configs4"] = {"value": n_dec / dt, "unit": "tok/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": "128-token prefill, then 32 greedy recurrent decode steps (Hyena modal state + FIR state + KV cache), "
                                      "batch 1, fp32", "ms_per_token": 1e3 * dt / n_dec}
        if engine is not None:
            # the engine on the same prompt through its cached path (prefill, eager first step, hipGraph replays), fed the ORACLE's
            # tokens so that one flipped argmax does not cascade: do the two agree on what they would have sampled?
            dev = engine.device
            c = engine.initialize_inference_params()
            e_logits = [engine(ids.to(dev), c)[0][0, -1].float().cpu()]
            for j in range(n_dec):
                c["mha"].seqlen_offset = c["hyena"].seqlen_offset = ids.shape[1] + j
                e_logits.append(engine(o_toks[j].to(dev), c)[0][0, -1].float().cpu())
            if hasattr(engine, "release_decode_graph"):
                engine.release_decode_graph()
            el, ol = torch.stack(e_logits), torch.stack(o_logits)
            legs["configs4"]["tokens_agree"] = float((el.argmax(-1) == ol.argmax(-1)).float().mean())
            legs["configs4"]["gpu_logits_rel_l2_same_tokens"] = float((el.double() - ol.double()).norm() / ol.double().norm())
            legs["configs4"]["tokens_compared"] = n_dec + 1
            # the floor those two numbers are read against: the SAME oracle class in its bf16 mode (a rounding after every eager op =
            # the arithmetic the reference itself runs in), on the same weights and the same tokens, executed by torch's eager GPU
            # kernels (checker only: nothing of libevo_mi355x.so).  The 32-block stack of random weights amplifies rounding noise, so
            # "within 1e-3 of fp32" is out of reach of ANY bf16 pipeline here; the engine should sit at or below this floor.
            try:
                ob = R.RefStripedHyena(cfg, {k: v for k, v in engine.state_dict().items()}, "bf16", device=dev)
                cb = ob.initialize_inference_params()
                b_logits = [ob(ids.to(dev), cb)[0][0, -1].float().cpu()]
                cb["mha"].seqlen_offset = cb["hyena"].seqlen_offset = ids.shape[1]
                for j in range(n_dec):
                    b_logits.append(ob(o_toks[j].to(dev), cb)[0][0, -1].float().cpu())
                    cb["mha"].seqlen_offset += 1
                    cb["hyena"].seqlen_offset += 1
                bl = torch.stack(b_logits)
                legs["configs4"]["eager_bf16_floor"] = {
                    "tokens_agree": float((bl.argmax(-1) == ol.argmax(-1)).float().mean()),
                    "logits_rel_l2_same_tokens": float((bl.double() - ol.double()).norm() / ol.double().norm()),
                    "what": "oracle in bf16 mode (the reference's eager bf16 arithmetic) vs the same fp32 oracle, same weights / prompt / tokens"}
                del ob, cb
            except Exception as e:  # noqa: BLE001
                legs["configs4"]["eager_bf16_floor"] = {"error": f"{type(e).__name__}: {e}"}
    return legs


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this script under torch.distributed.run,
    one per GPU, and pass their exit code on.  Fails loudly when the box has fewer than N devices."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("EVO_AMD_BENCH_SHARE_GPU", "0") != "1":
        sys.stderr.write(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible "
                         f"(torch.cuda.device_count()); refusing to benchmark fewer GPUs than asked for\n")
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--nt", type=int, default=8192)
    ap.add_argument("--skip-131k", action="store_true")
    ap.add_argument("--skip-ab", action="store_true", help="skip the in-process A/B legs (library_gemm_l3, mlp_gate_unfused, norm_unfused, attention_round4_kernel): "
                                                          "the profile runs want the headline step's kernels only")
    ap.add_argument("--skip-sp-predict", action="store_true", help="skip the stub-communicator rank of configs[3] (scaling_131k_predicted)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-box", action="store_true", help="skip the box-calibration probes (HBM copy, MFMA stream, library GEMM)")
    ap.add_argument("--skip-gen", action="store_true")
    ap.add_argument("--steps-131k", type=int, default=2)
    ap.add_argument("--sp-timeout", type=float, default=420.0, help="N > 1: seconds the sequence-parallel 131k leg may take")
    args = ap.parse_args()

    import torch.distributed as dist
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn_under_torchrun(args.gpus)                 # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)
    # EVO_AMD_BENCH_SHARE_GPU=1 (a SELF-TEST of the N > 1 code path on a one-GPU box, never a measurement): every rank runs
    # on cuda:0, the process group is gloo and the sequence-parallel leg exchanges through evo_amd.sp.HostStagedComm
    share_gpu = os.environ.get("EVO_AMD_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local = 0
    if torch.cuda.device_count() <= local:
        sys.stderr.write(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local}, {torch.cuda.device_count()} visible)\n")
        sys.exit(2)
    dist_on = world > 1
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))
        assert dist.get_world_size() == world
    n_gpus = world

    from evo_amd.ops import KernelTimer, default_ops
    ops = default_ops()                                   # raises if libevo_mi355x.so / the GPU is missing

    # ------------------------------------------------------------------ primary: 8 x 8,192 nt per GPU
    model = build_model("evo-1-8k-base", device)
    B, nt = args.batch, args.nt
    ids = acgt_ids(B, nt, 1234 + 1000 * rank, device)
    T = nt + 1
    step_stats = {}
    box = {}
    if not args.skip_box:
        try:
            box = box_probes(ops, device, local)
        except Exception as e:  # noqa: BLE001
            box = {"error": f"{type(e).__name__}: {e}"}
    with torch.inference_mode():
        with TelemetrySampler(local) as tel_head:
            dt = timed(lambda: scoring_step(model, ids), args.steps, args.warmup, dist_on, step_stats)
        box["headline_telemetry"] = tel_head.summary()
        # per-kernel HIP-event timings over a second, separately instrumented pass of the same steps
        ops.timer = KernelTimer()
        for _ in range(args.steps):
            scoring_step(model, ids)
        torch.cuda.synchronize()
        ksum = ops.timer.summary()
        ops.timer = None
    ms_per_step = dt / args.steps * 1e3
    value = n_gpus * B * nt / (dt / args.steps)

    D = 4096
    alg_bytes = B * T * (3 * D * 2 + D * 2)               # z in + y out per launch (32,768 B/token)
    roofline = hyena_roofline(ops, model, ksum, B, T, alg_bytes, device)
    attn_flops = B * 4 * D * T * T / 2                    # causal QK^T + PV per layer
    kernels = {k: {"launches_per_step": v[0] // args.steps, "avg_ms": v[1]} for k, v in ksum.items()}
    kernels["attn_fwd"]["tflops"] = attn_flops / (ksum["attn_fwd"][1] * 1e-3) / 1e12
    kernels["attn_fwd"]["mfma_frac"] = kernels["attn_fwd"]["tflops"] / MFMA_BF16_PEAK_TFLOPS
    # dense layers: "gemm" = hipBLASLt, "gemm_mfma" = the hand-written kernel (csrc/gemm.hip), "gemm_gate" = the same kernel with the
    # gated MLP's GELU * gate in its epilogue (l1 | l2 of every block; its launch also does the work of the former gelu_gate pass)
    gemm_ms = sum(ksum[k][1] * ksum[k][0] / args.steps for k in ("gemm", "gemm_mfma", "gemm_gate", "gemm_zt") if k in ksum)
    # the dense layers (87 % of the step) against the dense bf16 MFMA peak: 2 * M * N * K summed over the launches of a step
    dense_flop = 2.0 * B * T * 4096 * (12288 + 4096 + 22016 + 11008) * 32      # proj/Wqkv, out, l1|l2 (padded), l3 (padded K)
    roofline_dense = {"bound": "mfma", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "achieved": dense_flop / (gemm_ms * 1e-3) / 1e12, "frac": dense_flop / (gemm_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                      "kernels": "hipBLASLt (A/B knob) + gemmr_bf16_kernel" if "gemm" in ksum else "gemmr_bf16_kernel (csrc/gemm.hip)", "ms_per_step": gemm_ms,
                      "hand_written_share_of_dense_ms": sum(ksum[k][1] * ksum[k][0] / args.steps for k in ("gemm_mfma", "gemm_gate", "gemm_zt") if k in ksum) / gemm_ms,
                      "note": "2.5 PFLOP/s is the dense peak; the part is power-limited: both kernels run their MFMA pipes 82-86 % busy "
                              "at 1.6-1.7 GHz (profiles/r03_gemm_notes.txt)"}
    out = {
        "metric": "nucleotides/sec forward scoring, evo-1 7B", "value": value, "unit": "nt/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic ACGT, synthetic weights",
        "n_ranks": dist.get_world_size() if dist_on else 1,
        "collectives": (None if not dist_on else "SELF-TEST: gloo, host-staged, all ranks on one GPU (not a measurement)"
                        if dist.get_backend() == "gloo" else
                        "RCCL %s (torch.distributed backend nccl)" % ".".join(map(str, torch.cuda.nccl.version()))),
        "step_timing": step_stats,
        "config": {"workload": "evo-1-8k-base scoring, batch 8 x 8,192 nt per GPU (BASELINE configs[1])",
                   "batch_per_gpu": B, "nt": nt, "tokens_per_seq": T,
                   "parallelism": "independent batches per GPU" if n_gpus > 1 else "single GPU"},
        "model_tflops": flops_per_token(T) * B * T / (dt / args.steps) / 1e12,
        "roofline": roofline, "roofline_dense": roofline_dense, "kernels": kernels, "gemm_ms_per_step": gemm_ms,
        "gemm_library_launches_per_step": kernels.get("gemm", {}).get("launches_per_step", 0),
        "box": box,
        "weights_resident_GB": model.resident_bytes() / 1e9,
        "weights_resident_note": "the scoring model as this run used it (default: originals + norm-folded copies + operand tables); "
                                 "`generation.weights_resident_GB` is the same model on ONE weight set (fold_norms_)",
    }
    if "library_gemm_tflops" in box and "hbm_copy_GBs" in box:
        share = min(1.0, gemm_ms / (step_stats.get("hip_event_ms_median") or ms_per_step))
        cal = share * box["library_gemm_tflops"] / REF_GEMM_TFLOPS + (1.0 - share) * box["hbm_copy_GBs"] / REF_COPY_GBS
        box["dense_share_of_step"] = share
        box["calibration_factor"] = cal
        out["value_per_calibrated_box"] = value / cal
        out["value_per_calibrated_box_note"] = ("value / (dense_share * library_gemm_tflops / 1500 + (1 - dense_share) * hbm_copy_GBs / 6290): the headline "
                                                "on a box that gives the reference rates; compare THIS across rounds, `value` across code states on one box")
    # ------------------------------------------------------------------ the A/B legs' own reference: the DEFAULT routing timed exactly as the legs are
    # (3 steps behind one warm-up, here and again behind the last leg: the headline's 5 steps were timed minutes earlier in the process and the
    # part's clocks drift by ~1 % meanwhile -- compare a leg with `ab_reference`, not with the headline)
    def _ab_ref():
        with torch.inference_mode():
            dtr = timed(lambda: scoring_step(model, ids), 3, 1, dist_on)
        return dtr / 3 * 1e3
    if n_gpus == 1 and not args.skip_ab:
        try:
            out["ab_reference"] = {"ms_per_step_before_legs": _ab_ref(), "steps": 3,
                                   "note": "the default routing (this run's headline configuration) timed the way the A/B legs are, in front of and behind them"}
        except Exception as e:  # noqa: BLE001
            out["ab_reference"] = {"error": f"{type(e).__name__}: {e}"}
    # ------------------------------------------------------------------ the same step with the plain dense layers on hipBLASLt
    if n_gpus == 1 and ops.all_gemm_mfma and not args.skip_ab:
        try:
            ops.all_gemm_mfma = False
            with torch.inference_mode():
                dt2 = timed(lambda: scoring_step(model, ids), 3, 1, dist_on)
            out["library_gemm_l3"] = {"value": B * nt / (dt2 / 3), "unit": "nt/s", "ms_per_step": dt2 / 3 * 1e3, "steps": 3,
                                      "note": "l3 (32 launches, K = 11,008) on hipBLASLt through torch.addmm in the same process; the headline runs "
                                              "every dense layer on csrc/gemm.hip (the library is 1-3 % faster on that shape)"}
        except Exception as e:  # noqa: BLE001
            out["library_gemm_l3"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.all_gemm_mfma = True
    # ------------------------------------------------------------------ the same step with the gated MLP unfused (the round-2 default)
    if n_gpus == 1 and getattr(ops, "mlp_gate_fused", False) and not args.skip_ab:
        try:
            ops.mlp_gate_fused = False
            with torch.inference_mode():
                dt3 = timed(lambda: scoring_step(model, ids), 3, 1, dist_on)
            out["mlp_gate_unfused"] = {"value": B * nt / (dt3 / 3), "unit": "nt/s", "ms_per_step": dt3 / 3 * 1e3, "steps": 3,
                                       "note": "l1 | l2 as a plain dense layer + the gate kernel (ops.mlp_gate_fused = False) in the same process: "
                                               "what the one-launch form with GELU * gate in the dense layer's epilogue buys"}
        except Exception as e:  # noqa: BLE001
            out["mlp_gate_unfused"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.mlp_gate_fused = True
    # ------------------------------------------------------------------ the same step with the 65 separate RMSNorm passes (rounds 1-4)
    if n_gpus == 1 and getattr(ops, "fuse_norm", False) and not args.skip_ab:
        try:
            ops.fuse_norm = False
            with torch.inference_mode():
                dt7 = timed(lambda: scoring_step(model, ids), 3, 1, dist_on)
                ops.timer = KernelTimer()
                scoring_step(model, ids)
                torch.cuda.synchronize()
                k7 = ops.timer.summary()
                ops.timer = None
            out["norm_unfused"] = {"value": B * nt / (dt7 / 3), "unit": "nt/s", "ms_per_step": dt7 / 3 * 1e3, "steps": 3,
                                   "rmsnorm_launches": k7.get("rmsnorm", (0, None))[0], "rmsnorm_avg_ms": k7.get("rmsnorm", (0, None))[1],
                                   "rmsnorm_launches_headline": kernels.get("rmsnorm", {}).get("launches_per_step"),
                                   "rms_finalize_launches_headline": kernels.get("rms_finalize", {}).get("launches_per_step"),
                                   "note": "ops.fuse_norm = False in the same process: every RMSNorm as its own pass over the stream (a normalised copy "
                                           "written and read back) instead of a row factor in the epilogue of the dense layer that consumes it, with the "
                                           "statistic from the epilogue of the dense layer that wrote the stream (csrc/gemm.hip NF)"}
        except Exception as e:  # noqa: BLE001
            out["norm_unfused"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.fuse_norm = True
            ops.timer = None
    # ------------------------------------------------------------------ the same step with every row's last token as a ragged tile of the operator
    if n_gpus == 1 and getattr(ops, "hyena_tail_split", False) and not args.skip_ab:
        try:
            ops.hyena_tail_split = False
            with torch.inference_mode():
                dt8 = timed(lambda: scoring_step(model, ids), 3, 1, dist_on)
                ops.timer = KernelTimer()
                scoring_step(model, ids)
                torch.cuda.synchronize()
                k8 = ops.timer.summary()
                ops.timer = None
            ms8 = k8.get("hyena_mfma", (0, None))[1]
            out["hyena_tail_in_operator"] = {"value": B * nt / (dt8 / 3), "unit": "nt/s", "ms_per_step": dt8 / 3 * 1e3, "steps": 3,
                                             "hyena_ct_avg_ms": ms8, "hyena_ct_avg_ms_headline": kernels.get("hyena_mfma", {}).get("avg_ms"),
                                             "hyena_ct_frac_of_8TBs": None if not ms8 else B * T * (3 * D * 2 + D * 2) / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "note": "ops.hyena_tail_split = False in the same process: the token behind the 8,192 main tokens of every row as a "
                                                     "ragged 17th tile of hyena_ct (one valid step at a full tile's issue time; its projection through the "
                                                     "weight-streaming launch) -- the rounds 4-5 form; the headline runs it through the fused single-token launch"}
        except Exception as e:  # noqa: BLE001
            out["hyena_tail_in_operator"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.hyena_tail_split = True
            ops.timer = None
    # ------------------------------------------------------------------ the same step with the round-2..4 attention kernel
    if n_gpus == 1 and getattr(ops, "attn_w64", False) and not args.skip_ab:
        try:
            ops.attn_w64 = False
            with torch.inference_mode():
                dt6 = timed(lambda: scoring_step(model, ids), 3, 1, dist_on)
                ops.timer = KernelTimer()
                scoring_step(model, ids)
                torch.cuda.synchronize()
                k6 = ops.timer.summary()
                ops.timer = None
            out["attention_round4_kernel"] = {"value": B * nt / (dt6 / 3), "unit": "nt/s", "ms_per_step": dt6 / 3 * 1e3, "steps": 3,
                                              "attn_fwd_avg_ms": k6.get("attn_fwd", (0, None))[1],
                                              "note": "the same process with attn_fwd_pipe_kernel (csrc/attn.hip: 8 waves x 32 query rows, two waves per "
                                                      "SIMD) instead of attn_fwd_w64_kernel (csrc/attn_w64.hip: 4 waves x 64 rows, one per SIMD, V^T pre-pass)"}
        except Exception as e:  # noqa: BLE001
            out["attention_round4_kernel"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.timer = None
            ops.attn_w64 = True
    if n_gpus == 1 and not args.skip_ab and "ms_per_step_before_legs" in out.get("ab_reference", {}):
        try:
            out["ab_reference"]["ms_per_step_after_legs"] = _ab_ref()
        except Exception as e:  # noqa: BLE001
            out["ab_reference"]["error"] = f"{type(e).__name__}: {e}"
    # ------------------------------------------------------------------ the headline once more, behind the legs (same routing, same steps)
    if n_gpus == 1 and not args.skip_ab:
        try:
            again = {}
            with torch.inference_mode():
                with TelemetrySampler(local) as tel2:
                    dt_again = timed(lambda: scoring_step(model, ids), args.steps, 1, dist_on, again)
            out["headline_after_legs"] = {"value": B * nt / (dt_again / args.steps), "ms_per_step": dt_again / args.steps * 1e3, "steps": args.steps,
                                          "step_timing": again, "telemetry": tel2.summary(),
                                          "note": "the headline configuration timed a second time after the A/B legs (minutes later in the process): the "
                                                  "spread between the two is this box's drift, not code"}
        except Exception as e:  # noqa: BLE001
            out["headline_after_legs"] = {"error": f"{type(e).__name__}: {e}"}
    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    if rank == 0 and n_gpus == 1 and not args.skip_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(model)
            with torch.inference_mode():                  # the same sequence on the GPU engine: scores must agree
                ids0 = acgt_ids(1, 512, 1234, device)
                lp0 = scoring_step(model, ids0)
            out["cpu_baseline"]["gpu_score_same_input"] = float(lp0.double().mean().item())
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    del model
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ secondary: 131,072-nt context
    if not args.skip_131k:
        # N > 1: the sequence-parallel leg is the only part of this script with data-path collectives.  A rank that fails
        # inside it would leave the others waiting in RCCL until the collective timeout, and the headline line (measured above)
        # would never be printed: a watchdog prints what has been measured and ends every rank instead.
        watchdog = None
        if dist_on:
            import threading

            def _bail():
                if rank == 0:
                    out["ctx131k"] = {"error": f"sequence-parallel leg did not finish within {args.sp_timeout} s; abandoned"}
                    print(json.dumps(out), flush=True)
                os._exit(0)

            watchdog = threading.Timer(args.sp_timeout, _bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            out["ctx131k"] = bench_131k(args, device, rank, world, dist_on, ops)
        except Exception as e:  # noqa: BLE001  (report, never hide)
            out["ctx131k"] = {"error": f"{type(e).__name__}: {e}"}
            if dist_on:                                   # the other ranks may be inside a collective this rank left
                if rank == 0:
                    print(json.dumps(out), flush=True)
                    os._exit(0)
                os._exit(0)
        finally:
            if watchdog is not None:
                watchdog.cancel()

    # ------------------------------------------------------------------ generation (BASELINE configs[4]), N = 1 only
    if n_gpus == 1 and not args.skip_gen:
        try:
            out["generation"] = bench_generation(device)
        except Exception as e:  # noqa: BLE001
            out["generation"] = {"error": f"{type(e).__name__}: {e}"}

    if n_gpus > 1 and "ctx131k" in out and "error" not in out["ctx131k"]:
        out["scaling_131k"] = out["ctx131k"].get("scaling")   # BASELINE configs[3]: the sequence-split result, top level
    if n_gpus == 1 and isinstance(out.get("ctx131k"), dict) and "scaling_131k_predicted" in out["ctx131k"]:
        out["scaling_131k_predicted"] = out["ctx131k"].pop("scaling_131k_predicted")
    # the box's rates once more, compact: inside `config` (a key the driver's record keeps whole) and as the LAST key of the line (its stdout tail)
    bs = {k: out["box"].get(k) for k in ("hbm_copy_GBs", "mfma_probe_tflops", "library_gemm_tflops", "calibration_factor", "dense_share_of_step")
          if isinstance(out.get("box"), dict) and k in out["box"]}
    if bs:
        ht = out["box"].get("headline_telemetry", {})
        bs.update({"sclk_MHz_mean_headline": ht.get("sclk_MHz_mean"), "power_W_mean_headline": ht.get("power_W_mean"), "power_cap_W": ht.get("power_cap_W"),
                   "value_per_calibrated_box": out.get("value_per_calibrated_box"),
                   "headline_after_legs_value": (out.get("headline_after_legs") or {}).get("value")})
        out["config"]["box"] = bs
        out["box_summary"] = bs
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


def bench_generation(device, prompt=8192, new=1024):
    """BASELINE configs[4]: evo-1-131k-base, 8,192-nt prompt -> 1,024 new tokens, batch 1: full-prompt parallel prefill
    (exact carried Hyena state + KV cache), then the recurrent decode step (hipGraph-captured, weight-streaming
    GEMV).  Greedy (top_k = 1, deterministic) is the headline; the reference CLI's sampling profile (top_k = 4,
    temperature 1.0 [REF scripts/generate.py:28-30]) is timed beside it."""
    from evo_amd.generation import Generator
    from evo_amd.tokenizer import CharLevelTokenizer
    model = build_model("evo-1-131k-base", device)
    # round 6: the generation leg runs on ONE weight set (StripedHyena.fold_norms_: norm scales folded into the weights in place, no derived
    # copies) -- prefill, decode and the pool read the same 12.9 GB + 1.6 GB of operand tables
    gb_two = None
    try:
        with torch.inference_mode(False):
            model.prepare()
            gb_two = model.resident_bytes() / 1e9
            model.fold_norms_()
    except Exception as e:  # noqa: BLE001
        sys.stderr.write(f"bench.py: fold_norms_ failed ({type(e).__name__}: {e}); generation leg on the default weight copies\n")
    ids = acgt_ids(1, prompt, 777, device)[:, 1:]                    # no BOS: generate()'s default

    def run(g, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.generate(device=device, input_ids=ids, num_tokens=n, cached_generation=True, print_generation=False,
                   stop_at_eos=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    g1 = Generator(model, CharLevelTokenizer(512), top_k=1, top_p=1.0, temperature=1.0)
    run(g1, 3)
    t_pre = min(run(g1, 1) for _ in range(2))
    t_all = run(g1, 1 + new)
    dec = (t_all - t_pre) / new
    g4 = Generator(model, CharLevelTokenizer(512), top_k=4, top_p=1.0, temperature=1.0)
    run(g4, 3)
    dec4 = (run(g4, 1 + new) - t_pre) / new
    res = {"config": {"workload": f"evo-1-131k-base generation, {prompt}-nt prompt -> {new} new tokens, batch 1, greedy "
                                  f"(BASELINE configs[4])"},
           "prefill_ms": t_pre * 1e3, "prefill_nt_per_s": prompt / t_pre, "decode_ms_per_token": dec * 1e3,
           "decode_tokens_per_s": 1.0 / dec, "end_to_end_s": t_all,
           "weight_stream_GBps": 12.906 / dec, "hbm_frac": 12.906e9 / dec / 1e9 / HBM_PEAK_GBS,
           # decode is HBM-bound on the weights (12.9 GB/step) PLUS the KV read -- 3 attention layers x keys x
           # 2 (K, V) x 4096 x 2 B; averaged over the generated positions
           "kv_read_GB_per_token_avg": 3 * (prompt + new / 2) * 2 * 4096 * 2 / 1e9,
           "hbm_frac_with_kv": (12.906e9 + 3 * (prompt + new / 2) * 2 * 4096 * 2) / dec / 1e9 / HBM_PEAK_GBS,
           "top_k4_decode_ms_per_token": dec4 * 1e3,
           "graph_engaged": getattr(model, "decode_graph_replays", 0) > 0,
           "weights_resident_GB": model.resident_bytes() / 1e9, "weights_resident_GB_default_two_copies": gb_two,
           "one_weight_set": bool(getattr(model, "_norms_folded", False))}
    # ---- the semantic_design usage profile on the same 7B weights: many prompts x samples, continuous batching (evo_amd/pool.py)
    # [REF semantic_design/semantic_design.py:271-360].  Tokens/s INCLUDES the prompts' prefills; a pooled step streams the weights
    # once for all live slots, so the HBM fraction is weights x steps / time.
    try:
        from evo_amd.pool import DecodePool
        rng = np.random.default_rng(4242)
        pool_res = {}
        for n_slots, n_prompts, n_samp in ((8, 8, 2), (32, 16, 4)):
            prompts = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(512, 1025, size=n_prompts)]
            pool = DecodePool(model, CharLevelTokenizer(512), n_slots=n_slots, top_k=4, top_p=1.0, temperature=0.7, device=device)
            torch.manual_seed(0)
            pool.generate(prompts[:2], n_tokens=8, n_sample_per_prompt=n_samp)            # warm-up: graph capture at this slot count
            pool.stats = {"steps": 0, "prefills": 0, "tokens": 0}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            seqs, _, _ = pool.generate(prompts, n_tokens=128, n_sample_per_prompt=n_samp)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t0
            n_tok = sum(len(x) for x in seqs)
            pool_res[f"slots{n_slots}"] = {"tokens_per_s": n_tok / dtp, "generations": len(seqs), "new_tokens_each": 128,
                                           "prompts": n_prompts, "prompt_nt": "512-1024", "pooled_steps": pool.stats["steps"],
                                           "prefills": pool.stats["prefills"], "seconds": dtp,
                                           "ms_per_pooled_step_incl_prefills": dtp / max(1, pool.stats["steps"]) * 1e3,
                                           "hbm_frac_weights": 12.906e9 * pool.stats["steps"] / dtp / 1e9 / HBM_PEAK_GBS}
            if hasattr(model, "release_decode_graph"):
                model.release_decode_graph()
            del pool
        pool_res["single_stream_tokens_per_s"] = 1.0 / dec4
        res["pool"] = pool_res
    except Exception as e:  # noqa: BLE001
        res["pool"] = {"error": f"{type(e).__name__}: {e}"}
    del model
    torch.cuda.empty_cache()
    return res


def bench_131k(args, device, rank, world, dist_on, ops):
    """BASELINE configs[2] (world 1: batch 1 x 131,072 nt on one GPU) / configs[3] (world N: batch N, the SEQUENCE
    dimension sharded over the N ranks, evo_amd/sp.py).  For N > 1 every rank also times the single-GPU batch-1 pass
    in the same job, so the speed-up of the sequence split is an in-run ratio of nt/s."""
    import torch.distributed as dist
    from evo_amd.ops import KernelTimer
    model = build_model("evo-1-131k-base", device)
    nt = 131072
    T = nt + 1
    ids1 = acgt_ids(1, nt, 4321, device)
    single = lambda: scoring_step(model, ids1)            # noqa: E731
    scaling = None
    if world == 1:
        fn, B, par = single, 1, "single GPU"
    else:
        from evo_amd.sp import SequenceParallelScorer
        B = world
        ids = acgt_ids(B, nt, 4321, device)               # every rank builds the same batch, keeps its shard
        comm = None
        if dist.get_backend() == "gloo":                  # (EVO_AMD_BENCH_SHARE_GPU self-test)
            from evo_amd.sp import HostStagedComm
            comm = HostStagedComm()
        scorer = SequenceParallelScorer(model, rank, world, comm=comm)
        fn = lambda: scorer.score_logprobs(ids)           # noqa: E731
        par = f"sequence-parallel over {world} ranks (RCCL: neighbour halo send/recv, end-state all-gather, " \
              f"head<->sequence all-to-all)"
    st = {}
    with torch.inference_mode():
        dt = timed(fn, args.steps_131k, 1, dist_on, st)
        ops.timer = KernelTimer()
        fn()
        torch.cuda.synchronize()
        ks = ops.timer.summary()
        ops.timer = None
        if world > 1:
            dt1 = timed(single, args.steps_131k, 1, dist_on)              # the same job's 1-GPU reference (replicas)
            scorer.comm_profile = True                                    # raw duration of every exchange, serialised
            fn()
            torch.cuda.synchronize()
            comm = scorer.comm_summary()
            scorer.comm_profile = False
            per1 = dt1 / args.steps_131k
            per_layer = {k: {"count_per_step": v[0], "mean_ms": v[1]} for k, v in comm.items()}
            scaling = {"workload": f"evo-1-131k-base scoring, batch {B} x 131,072 nt, sequence split over {world} ranks "
                                   f"(BASELINE configs[3] at N = 8)",
                       "n_ranks": dist.get_world_size(), "value": B * nt / (dt / args.steps_131k), "unit": "nt/s",
                       "single_gpu_value_same_job": nt / per1, "speedup_vs_single_gpu": (B * nt / (dt / args.steps_131k)) / (nt / per1),
                       "collectives_serialised_ms": per_layer,
                       "collective_ms_per_step_serialised": sum(v[0] * v[1] for v in comm.values())}
    per = dt / args.steps_131k
    D = 4096
    Tl = T if world == 1 else (T + world - 1) // world
    alg_bytes = B * Tl * (3 * D * 2 + D * 2)
    if world == 1:
        roof = hyena_roofline(ops, model, ks, B, T, alg_bytes, device)
    elif "hyena_mfma" in ks:
        # sequence shards: stage 1 = the state-only walk of the single-pass kernel (reads z), the end states travel, stage 2 =
        # the full pass seeded with the carried state (reads z, writes y) -- launched once per row group and layer
        n2, ms2 = ks["hyena_mfma"]
        n1, ms1 = ks.get("hyena_mfma_state", (0, 0.0))
        per_layer = max(1, n2 // 29)
        alg_bytes = alg_bytes // per_layer
        roof = {"kernel": "hyena_ct_kernel (stage 2 of a shard: the full pass seeded with the carried state)", "bound": "hbm",
                "achieved": alg_bytes / (ms2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms2, "launches_per_layer": per_layer,
                "stage1_state_only_ms": ms1, "operator_2_launch_ms": ms1 + ms2,
                "operator_frac": alg_bytes / ((ms1 + ms2) * 1e-3) / 1e9 / HBM_PEAK_GBS}
    elif "hyena_apply" in ks:                             # (modal kernels: shards below the single-pass kernel's floor)
        apply_ms = ks["hyena_apply"][1]
        op_ms = apply_ms + ks["hyena_seg_state"][1] + ks["hyena_carry_scan"][1]
        alg_bytes = alg_bytes * 29 // ks["hyena_apply"][0]          # the row-group pipeline launches the operator per group
        roof = {"kernel": "hyena_apply_kernel", "bound": "hbm",
                "achieved": alg_bytes / (apply_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (apply_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": apply_ms, "operator_3_launch_ms": op_ms,
                "operator_frac": alg_bytes / (op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    else:
        roof = None
    res = {"value": B * nt / per, "unit": "nt/s", "ms_per_step": per * 1e3, "steps": args.steps_131k,
           "config": {"workload": f"evo-1-131k-base scoring, batch {B} x 131,072 nt", "parallelism": par},
           "step_timing": st,
           "model_tflops": flops_per_token(T) * B * T / per / 1e12,
           "roofline": roof,
           "kernels": {k: {"launches": v[0], "avg_ms": v[1]} for k, v in ks.items()}}
    if scaling is not None:
        res["scaling"] = scaling
    if world == 1 and not getattr(args, "skip_sp_predict", False):
        # BASELINE configs[3] cannot run on this pool (one GPU per box): one rank's kernels behind a stub communicator give the
        # per-rank compute time and, with stated link assumptions, a predicted 8-GPU rate (tools/sp_predict.py)
        try:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            from sp_predict import sp_predict
            res["scaling_131k_predicted"] = sp_predict(model, device, ops, single_ms=per * 1e3, acgt_ids=acgt_ids,
                                                       scoring_step=scoring_step)
        except Exception as e:  # noqa: BLE001
            res["scaling_131k_predicted"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and "attn_fwd" in ks and getattr(ops, "attn_w64", False) and not getattr(args, "skip_ab", False):
        try:                                              # the same 131k step on the round-2..4 attention kernel (one step)
            ops.attn_w64 = False
            with torch.inference_mode():
                ops.timer = KernelTimer()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                dt_old = time.perf_counter() - t0
                k_old = ops.timer.summary()
            res["attention_round4_kernel"] = {"ms_per_step": dt_old * 1e3, "attn_fwd_avg_ms": k_old["attn_fwd"][1],
                                              "attn_fwd_avg_ms_this_round": ks["attn_fwd"][1]}
        except Exception as e:  # noqa: BLE001
            res["attention_round4_kernel"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.timer = None
            ops.attn_w64 = True
    if world == 1 and getattr(ops, "fuse_norm", False) and not getattr(args, "skip_ab", False):
        try:                                              # the same 131k step with the 65 separate RMSNorm passes (two steps, the second timed)
            ops.fuse_norm = False
            with torch.inference_mode():
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                res["norm_unfused"] = {"ms_per_step": (time.perf_counter() - t0) * 1e3}
        except Exception as e:  # noqa: BLE001
            res["norm_unfused"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.fuse_norm = True
    if world == 1 and "attn_fwd" in ks:
        fl = 4 * D * T * T / 2
        res["kernels"]["attn_fwd"]["tflops"] = fl / (ks["attn_fwd"][1] * 1e-3) / 1e12
        res["kernels"]["attn_fwd"]["mfma_frac"] = res["kernels"]["attn_fwd"]["tflops"] / MFMA_BF16_PEAK_TFLOPS
    del model
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
