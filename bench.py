#!/usr/bin/env python
"""bench.py -- the headline benchmark: images/sec of the FULL YOLOv5m train step at 640x640.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = forward + build-targets + ComputeLoss + backward + clip_grad_norm_(10) + Adam(L2) on one
synthetic batch of 64 images per GPU (BASELINE.json configs[2]/[3]; weak scaling), bf16 activations /
weights with f32 accumulation, f32 master weights and optimizer state, random-init weights, inputs
resident in HBM before the timed region. Prints ONE JSON line (rank 0).

Extra legs on rank 0 (outside the timed region):
  * roofline: an eager pass with a HIP-event pair around every launch (events on the launch stream)
    gives the conv implicit-GEMM family's total time; achieved = algorithmic FLOPs (48.872
    GFLOP/img forward, the same again minus the stem for the data gradient) / that time, vs the dense
    bf16 MFMA peak (2.5 PFLOP/s). The same command under `rocprofv3 --kernel-trace --stats` is committed
    under profiles/.
  * cpu_baseline ("port"): the CPU oracle's train step (torch fp32, host cores) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s (this box's device copy measures 5.3-5.5 TB/s)
FWD_GFLOP_PER_IMAGE_640 = 48.872


def pmc_traffic(kernel=None):
    """(HBM bytes per launch, where the number comes from) for the kernel the roofline headline names (all template
    instances of that kernel, e.g. every wgrad_kernel<...> launch of the step; the PMC summary is keyed by the bare kernel
    name), or averaged over the conv kernels when no kernel is given / the summary does not hold it. bench.py cannot run
    rocprofv3 on itself inside its timed process, so this is an OFFLINE figure: the last committed PMC summary of the same
    workload (profiles/rNN_pmc_bench.json from tools/pmc_bench.sh: separate FETCH_SIZE / WRITE_SIZE passes, KiB units,
    gfx950 x2 on fetch, B=64 @ 640x640, --no-graph, first-touch fills included). (None, reason) when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench.json")))
    if not files:
        return None, "no profiles/r*_pmc_bench.json committed"
    rel = os.path.relpath(files[-1], ROOT)
    try:
        k = json.load(open(files[-1]))["kernels"]
        base = (kernel or "").split("<")[0].strip()
        if base in k:
            return round(k[base]["hbm_bytes_per_launch"]), f"offline: {rel} (all {base} launches of the step)"
        tot = n = 0.0
        for name in ("conv_igemm_kernel", "conv_pw_kernel", "conv_halo_kernel", "conv_igemm_multi_kernel"):
            if name in k:
                tot += k[name]["hbm_bytes_per_launch"] * k[name]["launches"]
                n += k[name]["launches"]
        return (round(tot / n) if n else None), f"offline: {rel} (average over the conv kernels' launches)"
    except Exception as e:
        return None, f"unreadable {files[-1]}: {e}"


def cpu_baseline(B=4, size=640, steps=3, world=1):
    """CPU leg: the oracle restatement of the reference path (model fwd, ComputeLoss, autograd bwd,
    clip, Adam) on the host cores. kind = "port". Bounded: B=4, 1 warm-up + `steps` timed steps.
    world > 1: torch.distributed.run starts every rank with OMP_NUM_THREADS=1; rank 0 takes its share of the cores back."""
    from oracle import loss_ref, model_ref
    if world > 1:
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    from yolov5m_amd.utils.synth import synth_state_dict, synth_images, synth_labels
    torch.manual_seed(0)
    sd = synth_state_dict()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and "running" not in k and "anchors" not in k}
    full = dict(sd)
    full.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=5e-4, weight_decay=5e-4)
    x = synth_images(B, size, size)
    t = synth_labels(B, 8)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        out = model_ref.forward(full, x, training=True, new_stats={})
        loss, _ = loss_ref.compute_loss_ultra(out, t, sd["head.anchors"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 10.0)
        opt.step()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    med = sorted(times)[len(times) // 2]                   # 3 timed steps: the true median
    return {"value": round(B / med, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed train steps (+1 warm-up) of B={B} @ {size}x{size}, torch fp32 on the host "
                      f"({os.cpu_count()} logical CPUs)"}


def first_loss_check(first_loss, B, S, rank):
    """the first step's ComputeLoss next to the REAL reference's f32 value for the same inputs and initial weights
    (tests/golden/g13_precision.npz, made by tests/golden/make_golden.py from /root/reference; B = 64 @ 640x640, rank 0)"""
    out = {"first_step_loss": round(first_loss, 4)}
    try:
        import numpy as np
        if B == 64 and S == 640 and rank == 0:
            ref = float(np.load(os.path.join(ROOT, "tests", "golden", "g13_precision.npz"))["b64_640/loss"])
            out["first_step_loss_reference_f32"] = round(ref, 4)
            out["first_step_loss_rel_err"] = float(f"{abs(first_loss - ref) / ref:.2e}")
    except Exception:
        pass
    return out


def forward_leg(model, dev, B=32, size=640, iters=10, warm=3):
    """configs[1]: forward only, batch 32 @ 640x640, bf16: eval mode (BatchNorm folded into the conv epilogues) and
    train mode (batch statistics), images/s each (inputs resident in HBM)."""
    from yolov5m_amd.utils.synth import synth_images
    x = synth_images(B, size, size, seed="img/fwd").to(dev)
    out = {}
    for mode in ("eval", "train"):
        model.train(mode == "train")
        with torch.no_grad():
            for _ in range(warm):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                model(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        out[mode] = {"images_per_sec": round(B / dt, 1), "ms": round(dt * 1e3, 3)}
        # the forward's own roofline: its algorithmic HBM bytes (every operand of every launch once, Engine.algorithmic_bytes) and its
        # FLOPs against the time -- at these shapes the bf16 forward is BANDWIDTH-bound (B=32 @ 640 eval: 7.5 GB = 1.5 ms at 5 TB/s
        # against 0.63 ms of MFMA work at the dense peak), so the HBM fraction is the one to read
        try:
            eb = model._engine_for(x).algorithmic_bytes()["forward"]
            gb = (eb["act_read"] + eb["act_written"] + eb["par_read"] + eb["par_written"]) / 1e9
            fl = FWD_GFLOP_PER_IMAGE_640 * B * (size / 640.0) ** 2
            out[mode].update(algorithmic_GB=round(gb, 3), GBps=round(gb / dt, 1), frac_of_hbm_peak=round(gb / dt / PEAK_HBM_GBPS, 4),
                             TFLOPs=round(fl / dt / 1e3, 1), frac_of_mfma_peak=round(fl / dt / 1e3 / PEAK_BF16_TFLOPS, 4))
        except Exception as e:  # noqa: BLE001
            out[mode]["roofline_error"] = f"{type(e).__name__}: {e}"
    model.train(True)
    tag = " (BASELINE.json configs[1])" if (B, size) == (32, 640) else ""
    return {"workload": f"forward only, batch {B} @ {size}x{size}, bf16{tag}", "unit": "images/s", **out}


def detect_leg(dev, model=None, B=128, size=1280, iters=5, warm=2, model_iters=3):
    """configs[4]: decode + per-image NMS on B x N candidate boxes (N = 100 800 at 1280^2), synthetic
    logits regime (ii) of SURVEY 8d: obj-logit ~ N(-5, 2^2), box/cls logits ~ N(0,1), seed 0.
    boxes/sec = B*N candidates consumed by decode + threshold + NMS per second (inputs resident in HBM)."""
    from yolov5m_amd import config
    from yolov5m_amd.utils.plot_utils import cells_to_bboxes
    from yolov5m_amd.utils.bboxes_utils import nms_batched
    g = torch.Generator(device=dev).manual_seed(0)
    logits = []
    for s_ in (8, 16, 32):
        ny = nx = size // s_
        t = torch.randn((B, 3, ny, nx, 85), generator=g, device=dev, dtype=torch.float32)
        t[..., 4] = t[..., 4] * 2.0 - 5.0
        logits.append(t)
    anchors = (torch.tensor(config.ANCHORS).float().view(3, -1, 2) / torch.tensor([8., 16., 32.]).view(3, 1, 1)).to(dev)
    N = sum(t.shape[1] * t.shape[2] * t.shape[3] for t in logits)
    out = {}
    regimes = {"detect.py (0.25, 0.45)": (0.25, 0.45), "eval (0.01, 0.6)": (0.01, 0.6),
               "worst case: obj ~ N(0,1), all-pass regime (i), (0.25, 0.45)": (0.25, 0.45)}
    for name, (thr, iou) in regimes.items():
        if name.startswith("worst"):
            for t in logits:                       # regime (i) of SURVEY 8d: sigmoid(obj) ~ 0.5, (nearly) every box passes
                t[..., 4] = (t[..., 4] + 5.0) * 0.5
        for _ in range(warm):
            boxes = cells_to_bboxes(logits, anchors, [8, 16, 32], is_pred=True, to_list=False)
            rows, idx, cnt = nms_batched(boxes, iou, thr, 300)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        td = tn = 0.0
        for _ in range(iters):
            e0.record()
            boxes = cells_to_bboxes(logits, anchors, [8, 16, 32], is_pred=True, to_list=False)
            e1.record()
            rows, idx, cnt = nms_batched(boxes, iou, thr, 300)
            e2.record()
            torch.cuda.synchronize()
            td += e0.elapsed_time(e1)
            tn += e1.elapsed_time(e2)
        td, tn = td / iters, tn / iters
        cand = int((boxes[..., 1] > thr).sum())
        out[name] = {"boxes_per_sec": round(B * N / ((td + tn) * 1e-3)), "decode_ms": round(td, 3), "nms_ms": round(tn, 3),
                     "candidates_per_image": round(cand / B, 1), "kept_per_image": round(float(cnt.float().mean()), 1),
                     "decode_GBps": round(B * N * (85 * 4 + 24) / (td * 1e-3) / 1e9, 1)}
    if model is not None:
        # the whole detect.py flow (reference detect.py:50-54) with the MODEL's logits: eval forward at batch B @ size^2
        # (bf16, BatchNorm folded; the second conv's 5 GB input view runs in slabs of whole images), decode, NMS.
        # Random-init weights = regime (i): nearly every candidate passes the threshold.
        del logits
        torch.cuda.empty_cache()
        model.eval()
        x = torch.rand((B, 3, size, size), generator=g, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for _ in range(warm):
                o = model(x)
                boxes = cells_to_bboxes(o, anchors, [8, 16, 32], is_pred=True, to_list=False)
                rows, idx, cnt = nms_batched(boxes, 0.45, 0.25, 300)
            torch.cuda.synchronize()
            tf = td = tn = 0.0
            for _ in range(model_iters):
                e0.record()
                o = model(x)
                e1.record()
                boxes = cells_to_bboxes(o, anchors, [8, 16, 32], is_pred=True, to_list=False)
                e2.record()
                rows, idx, cnt = nms_batched(boxes, 0.45, 0.25, 300)
                e3 = torch.cuda.Event(enable_timing=True)
                e3.record()
                torch.cuda.synchronize()
                tf += e0.elapsed_time(e1) / model_iters
                td += e1.elapsed_time(e2) / model_iters
                tn += e2.elapsed_time(e3) / model_iters
        gb1280 = None
        try:
            eb = model._engine_for(x).algorithmic_bytes()["forward"]
            gb1280 = (eb["act_read"] + eb["act_written"] + eb["par_read"] + eb["par_written"]) / 1e9
        except Exception:  # noqa: BLE001
            pass
        model.train(True)
        model._engines = {}
        flops = B * FWD_GFLOP_PER_IMAGE_640 * (size / 640.0) ** 2
        out["forward_1280"] = {"workload": f"model forward (eval, bf16) + decode + NMS (0.25, 0.45), batch {B} @ {size}x{size}, "
                                           "random-init weights", "forward_ms": round(tf, 2),
                               "forward_images_per_sec": round(B / (tf * 1e-3), 1),
                               "forward_TFLOPs": round(flops / tf, 1),
                               # (bandwidth-bound: 119 GB of algorithmic traffic = 24 ms at 5 TB/s against 10 ms of MFMA work)
                               "forward_algorithmic_GB": round(gb1280, 2) if gb1280 else None,
                               "forward_GBps": round(gb1280 / (tf * 1e-3), 1) if gb1280 else None,
                               "forward_frac_of_hbm_peak": round(gb1280 / (tf * 1e-3) / PEAK_HBM_GBPS, 4) if gb1280 else None,
                               "decode_ms": round(td, 3), "nms_ms": round(tn, 3),
                               "end_to_end_images_per_sec": round(B / ((tf + td + tn) * 1e-3), 1),
                               "end_to_end_boxes_per_sec": round(B * N / ((tf + td + tn) * 1e-3)),
                               "kept_per_image": round(float(cnt.float().mean()), 1)}
    return {"workload": f"decode + NMS, batch {B} @ {size}x{size} ({N} candidate boxes/image), synthetic logits "
                        f"(obj ~ N(-5,2^2)), max_detections 300" + (" (BASELINE.json configs[4])" if (B, size) == (128, 1280) else ""),
            "unit": "boxes/s", **out}


def _guarded(name, fn):
    """an untimed leg must never cost the JSON line: a Python error in it becomes {"error": ...} (traceback on stderr)"""
    try:
        return fn()
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        return {"error": f"{name} leg failed: {type(e).__name__}: {str(e)[:300]}"}


def roofline_leg(step, model, images, targets, dtype, passes=2):
    """rank 0, outside the timed region: HIP-event pairs around every launch of the step (profile_step) -> the `roofline` object"""
    fams = {}
    cls = {"spatial": [0.0, 0.0, 0.0, 0], "pointwise": [0.0, 0.0, 0.0, 0], "fused_pw_bwd": [0.0, 0.0, 0.0, 0]}     # [ms, flop, bytes, launches]
    esz = 2 if dtype == "bf16" else 4
    kernels, insitu = {}, {}
    for _ in range(passes):
        # the step's OWN schedule (weight gradients on the forked stream next to the main stream's kernels), every
        # launch timed by HIP events on the stream it runs on: what a launch costs inside the step
        _f, _c, kern = step.profile_step(images, targets, detail="kernels", overlapped=True)
        for name, ms_, fl, *by in kern:
            k = insitu.setdefault(name, [0.0, 0.0, 0, 0.0])
            k[0] += ms_; k[1] += fl; k[2] += 1; k[3] += by[0] if by else 0.0
    for _ in range(passes):
        # the same launches with the forked stream serialised: per-family times that add up
        fam, convs, kern = step.profile_step(images, targets, detail="kernels")
        for name, ms_, fl, *by in kern:
            k = kernels.setdefault(name, [0.0, 0.0, 0, 0.0])
            k[0] += ms_; k[1] += fl; k[2] += 1; k[3] += by[0] if by else 0.0
        for k, (m_, n_) in fam.items():
            a, b = fams.get(k, (0.0, 0))
            fams[k] = (a + m_, b + n_)
        for ms_, a in convs:
            c = cls["pointwise" if a.th * a.tw == 1 else "spatial"]
            c[0] += ms_
            c[1] += 2.0 * a.M * a.N * a.K
            c[2] += float(a.M) * (a.K / (a.th * a.tw) + a.N) * esz + float(a.N) * a.K * esz
            c[3] += 1
        for ms_, a in step.last_bwd_pw:
            # fused pointwise backward: algorithmic bytes = dz + y (N channels each) + x + dx (C channels each; + the
            # accumulation source when dx is added onto a tensor), 2 bytes per element
            c = cls["fused_pw_bwd"]
            c[0] += ms_
            c[1] += 4.0 * a.M * a.N * a.C
            c[2] += float(a.M) * (2 * a.N + (3 if (a.accumulate or a.res) else 2) * a.C) * esz
            c[3] += 1
    eng = model._engine_for(images)
    fwd_flops = eng.conv_flops()
    stem = eng.layers[0]
    dgrad_flops = fwd_flops - 2 * stem.M * stem.cout * stem.cin_real * stem.k * stem.k
    conv_ms, conv_n = fams.get("conv_igemm", (0.0, 1))
    conv_ms /= passes
    conv_n //= passes
    achieved = (fwd_flops + dgrad_flops) / (conv_ms * 1e-3) / 1e12
    wg_ms, wg_n = fams.get("wgrad", (0.0, 1))
    # headline = the ONE kernel instantiation with the largest share of the step, by summed launch time INSIDE the
    # step's own overlapped schedule (HIP events on the launching stream); `frac` is that in-situ figure, the
    # serialised one (forked stream run inline) sits beside it; the conv family and the per-kernel table follow
    ranked = sorted(insitu.items(), key=lambda kv: -kv[1][0])
    dom_name, (dom_ms, dom_fl, dom_n, dom_by) = ranked[0]
    ser_ms, ser_fl, ser_n, ser_by = kernels.get(dom_name, (dom_ms, dom_fl, dom_n, dom_by))
    hbm = dom_by > 0            # a kernel that carries algorithmic BYTES (the fused pointwise backward) is HBM-bound
    if hbm:
        dom_ach, ser_ach, peak, unit = dom_by / (dom_ms * 1e-3) / 1e9, ser_by / (ser_ms * 1e-3) / 1e9, PEAK_HBM_GBPS, "GB/s"
    else:
        dom_ach, ser_ach, peak, unit = dom_fl / (dom_ms * 1e-3) / 1e12, ser_fl / (ser_ms * 1e-3) / 1e12, PEAK_BF16_TFLOPS, "TFLOP/s"
    traffic, traffic_src = pmc_traffic(dom_name)

    def row(n_, v):
        r_ = {"kernel": n_, "ms_per_step": round(v[0] / passes, 3), "launches_per_step": v[2] // passes,
              "achieved_TFLOPs": round(v[1] / (v[0] * 1e-3) / 1e12, 1),
              "frac": round(v[1] / (v[0] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
              "ms_per_step_serialised": round(kernels.get(n_, v)[0] / passes, 3)}
        if v[3] > 0:
            r_.update(bound="hbm", achieved_GBps=round(v[3] / (v[0] * 1e-3) / 1e9, 1),
                      frac=round(v[3] / (v[0] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4))
        return r_
    return {
        "bound": "hbm" if hbm else "mfma", "kernel": dom_name,
        "achieved": round(dom_ach, 2), "peak": peak, "unit": unit,
        "frac": round(dom_ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
        "timing": "HIP events around every launch on its own stream, inside the step's overlapped schedule (eager pass)",
        "launches_per_step": dom_n // passes, "avg_launch_us": round(dom_ms * 1e3 / max(dom_n, 1), 2),
        "algorithmic_gflop_per_launch": round(dom_fl / max(dom_n, 1) / 1e9, 2),
        "algorithmic_MB_per_launch": round(dom_by / max(dom_n, 1) / 1e6, 2) if hbm else None,
        "share_of_step_ms": round(dom_ms / passes, 3),
        "achieved_serialised": round(ser_ach, 2), "frac_serialised": round(ser_ach / peak, 4),
        "avg_launch_us_serialised": round(ser_ms * 1e3 / max(ser_n, 1), 2),
        "by_kernel": [row(n_, v) for n_, v in ranked[:8]],
        "conv_family": {"what": "forward conv + data gradient, all y5m_conv launches of one step",
                        "achieved_TFLOPs": round(achieved, 2), "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                        "launches_per_step": conv_n, "avg_launch_us": round(conv_ms * 1e3 / max(conv_n, 1), 2),
                        "algorithmic_gflop_per_step": round((fwd_flops + dgrad_flops) / 1e9, 1)},
        "family_ms_per_step": {k: round(v[0] / passes, 3) for k, v in sorted(fams.items(), key=lambda kv: -kv[1][0])},
        "wgrad_tflops": round(fwd_flops / (wg_ms / passes * 1e-3) / 1e12, 2) if wg_ms else None,
        # the same launches split by what bounds them: k x k taps (MFMA) vs 1x1 (HBM: every input and output
        # element moves once, M*(Cin+Cout) elements + the weights)
        "by_class": {
            "spatial_convs(taps>1)": {"bound": "mfma", "launches_per_step": cls["spatial"][3] // passes,
                                      "ms_per_step": round(cls["spatial"][0] / passes, 3),
                                      "achieved_TFLOPs": round(cls["spatial"][1] / max(cls["spatial"][0], 1e-9) / 1e9, 1),
                                      "frac": round(cls["spatial"][1] / max(cls["spatial"][0], 1e-9) / 1e9 / PEAK_BF16_TFLOPS, 4)},
            "pointwise_convs(1x1)": {"bound": "hbm", "launches_per_step": cls["pointwise"][3] // passes,
                                     "ms_per_step": round(cls["pointwise"][0] / passes, 3),
                                     "achieved_GBps": round(cls["pointwise"][2] / max(cls["pointwise"][0], 1e-9) / 1e6, 1),
                                     "achieved_TFLOPs": round(cls["pointwise"][1] / max(cls["pointwise"][0], 1e-9) / 1e9, 1),
                                     "frac": round(cls["pointwise"][2] / max(cls["pointwise"][0], 1e-9) / 1e6 / PEAK_HBM_GBPS, 4)},
            "fused_pointwise_backward(bn apply + dgrad + wgrad)": {
                "bound": "hbm", "launches_per_step": cls["fused_pw_bwd"][3] // passes, "ms_per_step": round(cls["fused_pw_bwd"][0] / passes, 3),
                "achieved_GBps": round(cls["fused_pw_bwd"][2] / max(cls["fused_pw_bwd"][0], 1e-9) / 1e6, 1),
                "achieved_TFLOPs": round(cls["fused_pw_bwd"][1] / max(cls["fused_pw_bwd"][0], 1e-9) / 1e9, 1),
                "frac": round(cls["fused_pw_bwd"][2] / max(cls["fused_pw_bwd"][0], 1e-9) / 1e6 / PEAK_HBM_GBPS, 4)},
        },
    }


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """`python bench.py --gpus N` without torch.distributed.run in front: start the N ranks ourselves (one process per GPU,
    rendezvous on 127.0.0.1 and a free port) by re-executing this command line under torch.distributed.run. Returns the
    launcher's exit code. Refuses loudly when fewer than N devices are visible (unless Y5M_DIST_BACKEND=gloo, which folds
    the ranks onto the visible devices: the 1-GPU box's way to run the whole multi-process flow)."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("Y5M_DIST_BACKEND") != "gloo":
        sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) visible; refusing to report a {n}-GPU number\n")
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL / cross-process device memory on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--loss", default="ultralytics", choices=["ultralytics", "yolo"],
                    help="ultralytics = ComputeLoss (the BASELINE.json workload); yolo = YOLO_LOSS, the reference's default loss "
                         "(train.py:102-106) in the same fused step -- its line is labelled as NOT a BASELINE.json configuration")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-detect", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="one all-reduce after the backward pass instead of the bucketed, overlapped exchange")
    # sizes of the untimed legs (defaults = BASELINE.json configs[1] / configs[4] / the CPU sample; tests/emu/dry_run_bench.py runs
    # every leg at a tiny size on the CPU executor)
    ap.add_argument("--fwd-shape", default="32x640", help="forward leg: batch x size")
    ap.add_argument("--detect-shape", default="128x1280", help="detect leg: batch x size")
    ap.add_argument("--cpu-shape", default="4x640", help="cpu_baseline leg: batch x size")
    ap.add_argument("--leg-iters", type=int, default=0, help="iterations of the untimed legs (0 = their defaults; the dry run passes 1)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch.distributed as dist
    from yolov5m_amd import _lib, config, parallel
    from yolov5m_amd.model import YOLOV5m
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    from yolov5m_amd.utils.training_utils import NativeTrainStep
    from yolov5m_amd.utils.synth import synth_images, synth_labels

    rank, local, world = parallel.init_from_env()
    # rank 0 runs the untimed legs (roofline, cpu_baseline) alone while the other ranks wait: they wait on a gloo group with a
    # two-hour timeout, not in an RCCL barrier whose watchdog would tear the job down after the process group's default timeout
    exit_group = None
    if world > 1:
        import datetime
        exit_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=2))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the run")
    backend = dist.get_backend() if world > 1 else None
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s)")
    local = torch.cuda.current_device() if world > 1 else local      # (gloo on a smaller box: ranks folded onto the devices)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    _lib.check(_lib.lib().y5m_device_ok(), "y5m_device_ok")

    # who is running: one line per rank (device, PCI bus id, RCCL version, CU cap of the persistent kernels) gathered on rank 0;
    # with RCCL every rank must sit on its OWN device
    props = torch.cuda.get_device_properties(local)
    ident = {"rank": rank, "device_index": local, "name": props.name, "gcn_arch": getattr(props, "gcnArchName", None),
             "pci_bus_id": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}",
             "uuid": str(getattr(props, "uuid", "")), "persistent_cus": int(_lib.lib().y5m_persistent_cu_count())}
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if backend == "nccl":
            keys = {(d["uuid"], d["pci_bus_id"], d["device_index"]) for d in idents}
            if len(keys) != world:
                raise SystemExit(f"bench.py: {world} RCCL ranks on {len(keys)} distinct device(s): {idents}")
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl_version = None

    B, S = args.batch, args.size
    torch.manual_seed(0)                                   # identical random init on every rank
    model = YOLOV5m(first_out=config.FIRST_OUT, nc=80, anchors=config.ANCHORS,
                    ch=(config.FIRST_OUT * 4, config.FIRST_OUT * 8, config.FIRST_OUT * 16)).to(dev)
    model.compute_dtype = args.dtype
    model.train()
    model.flatten_parameters()
    parallel.broadcast_parameters(model)
    if args.loss == "yolo":
        from yolov5m_amd.loss import YOLO_LOSS
        loss_fn = YOLO_LOSS(model, rect_training=False)
    else:
        loss_fn = ComputeLoss(model)
    hook = parallel.GradAllReduce(world, timing=True) if world > 1 else None
    step = NativeTrainStep(model, loss_fn, nt_max=B * 8, use_graph=not args.no_graph, grad_hook=hook,
                           overlap=not args.no_overlap)

    # inputs resident in HBM before timing: the batch is written once into the step's static input buffer (what a
    # device-side loader / y5m_preprocess_u8 does), so step() makes no further copy of the images
    images = step.input_buffer(B, S, S)
    images.copy_(synth_images(B, S, S, seed=f"img/rank{rank}").to(dev))
    targets = synth_labels(B, 8, seed=f"lab/rank{rank}")
    if args.loss == "yolo":
        # the reference's collate_fn hands YOLO_LOSS host arrays (dataset.py:199-202): the step uploads them (8 boxes x 5 float64
        # per image) into its static buffers on every call, inside the timed region
        t = targets.numpy().astype("float64")
        targets = tuple(t[t[:, 0] == b][:, 1:] for b in range(B))
    else:
        targets = targets.to(dev)

    first_loss = None
    for _ in range(max(args.warmup, 1)):
        lo = step.step(images, targets)
        if first_loss is None:
            first_loss = float(lo[0])                      # the very first step: comparable with the reference's value
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lo = step.step(images, targets)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    final_loss = float(lo[0])
    # the exchange of the LAST timed step on every rank: per-bucket milliseconds (gradients final -> reduced) and the time
    # the compute stream waited for it before the optimizer; rank 0 reports the maximum over ranks
    xstats = hook.stats() if hook is not None else None
    if world > 1:
        nb = len(xstats["buckets"]) if xstats else 0
        v = torch.tensor(([b["ms"] for b in xstats["buckets"]] + [xstats["allreduce_exposed_ms"]]) if xstats else [0.0],
                         dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        if xstats:
            for b, ms_ in zip(xstats["buckets"], v[:nb].tolist()):
                b["ms"] = round(ms_, 3)
            xstats["allreduce_exposed_ms"] = round(float(v[nb]), 3)

    # the same workload with the PLAIN exchange (one all-reduce after the backward pass), 2 warm-up + 5 timed steps in this
    # same invocation, so that an N-GPU record carries overlapped and plain step times side by side
    plain = None
    if world > 1 and not args.no_overlap:
        hook2 = parallel.GradAllReduce(world, timing=True)          # (same event pairs as the overlapped leg)
        step2 = NativeTrainStep(model, loss_fn, nt_max=B * 8, use_graph=not args.no_graph, grad_hook=hook2, overlap=False)
        for _ in range(2):
            step2.step(images, targets)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step2.step(images, targets)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        tp = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        plain = {"steps": 5, "ms_per_step": round(float(tp[0]) / 5 * 1e3, 3)}
        del step2

    if rank != 0:
        if world > 1:
            dist.barrier(group=exit_group)
        return
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    out = {
        "metric": "images/sec (train step, 640x640)", "value": round(value, 2), "unit": "images/s",
        "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "dist_backend": backend,
        "rccl_version": rccl_version, "ranks": idents,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"YOLOv5m full train step (fwd + {'YOLO_LOSS' if args.loss == 'yolo' else 'ComputeLoss'} + bwd + clip + Adam), batch {B}/GPU @ "
                               f"{S}x{S}, random-init weights, 8 boxes/image"
                               + (" (NOT BASELINE.json configs[2]: --loss yolo = the reference's default YOLO_LOSS, whose in-place anchor "
                                  "decay leaves the anchors at zero after the first batches)" if args.loss == "yolo" else
                                  (" (BASELINE.json configs[2]" + ("/[3]" if world > 1 else "") + ")") if (B, S) == (64, 640)
                                  else " (NOT a BASELINE.json configuration: --batch / --size given)"),
                   "loss": args.loss, "global_batch": world * B, "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                   **(first_loss_check(first_loss, B, S, rank) if args.loss != "yolo" else {"first_step_loss": round(first_loss, 4)}),
                   "final_loss": round(final_loss, 4)},
    }
    r4 = int(_lib.lib().y5m_r4_kernel_forms())               # (as the library parsed Y5M_R4_KERNELS, once)
    out["kernel_forms"] = {"Y5M_R4_KERNELS": r4,
                           "note": "bit mask of the round-4 kernel rewrites in use (1 wgrad_rows, 2 bn_act, 4 bwd_stem, 8 bwd_pw, "
                                   "16 bn_bwd_reduce); 0 = the round-3 forms, the ones that passed the GPU suite on hardware"}
    if world > 1:
        out["exchange"] = {"what": "SUM all-reduce of the flat f32 gradient buffer (84.8 MB), "
                                   + ("one message after the backward pass" if args.no_overlap else
                                      "bucketed in backward order and overlapped with the backward pass (one captured "
                                      "graph per segment; buckets issued from a communication stream)"),
                           "last_step_max_over_ranks": xstats,
                           # (what the exposed time should be on xGMI)
                           "plain_exchange_same_run": plain,
                           # approximate: 5 plain steps against args.steps overlapped ones, a second NativeTrainStep (fresh Adam
                           # state, its own captured graphs) on the same model
                           "overlap_gain_ms_per_step_approx": round(plain["ms_per_step"] - ms, 3) if plain else None}
    def step_bytes():
        # the plan's algorithmic HBM bytes (every operand of every launch moved once: Engine.algorithmic_bytes, pinned by
        # tests/test_traffic_cpu.py) against the step time: the whole step's distance from the HBM roofline
        r = step.algorithmic_bytes(model._engine_for(images))
        rate = r["total_bytes"] / (ms * 1e-3) / 1e9
        top = sorted(r["by_kind"].items(), key=lambda kv: -kv[1][0])[:6]
        return {"algorithmic_GB_per_step": round(r["total_bytes"] / 1e9, 3), "GBps_at_step_time": round(rate, 1),
                "frac_of_hbm_peak": round(rate / PEAK_HBM_GBPS, 4), "GB_by_kind": {k: round(v[0] / 1e9, 3) for k, v in top}}
    out["step_bytes"] = _guarded("step_bytes", step_bytes)
    if not args.no_roofline:
        out["roofline"] = _guarded("roofline", lambda: roofline_leg(step, model, images, targets, args.dtype, args.leg_iters or 2))
    if world == 1 and not args.no_detect:
        del step, images
        model._engines = {}
        torch.cuda.empty_cache()
        fb, fs = (int(v) for v in args.fwd_shape.split("x"))
        db, ds = (int(v) for v in args.detect_shape.split("x"))
        li = args.leg_iters
        out["forward"] = _guarded("forward", lambda: forward_leg(model, dev, fb, fs, *((li, li) if li else ())))
        model._engines = {}
        torch.cuda.empty_cache()
        out["detect"] = _guarded("detect", lambda: detect_leg(dev, model, db, ds, *((li, li, li) if li else ())))
    if not args.no_cpu_baseline:
        # (rank 0's host cores; with N > 1 the other ranks wait at the exit barrier -- a gloo group with a two-hour timeout)
        cb, cs = (int(v) for v in args.cpu_shape.split("x"))
        out["cpu_baseline"] = _guarded("cpu_baseline", lambda: cpu_baseline(cb, cs, *((args.leg_iters,) if args.leg_iters else ()), world=world))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(group=exit_group)


if __name__ == "__main__":
    main()
