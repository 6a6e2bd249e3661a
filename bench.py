#!/usr/bin/env python3
"""Benchmark of the render-loss hot path on MI355X (contract: see DESIGN.md section "Measurement").

A step = one pass of the hot path over one synthetic batch resident in HBM:
  raw Gaussian-head output (B,P,23), drawn o ~ N(0,1) i.i.d. as SURVEY 8d's primary regime prescribes -> head activations -> batched HIP rasterizer forward over all B*V views ->
  focal-L2 render loss -> HIP rasterizer backward -> dL/d(head output).   This is what `value` times.
A second timed region (reported under "train_step_with_head", not `value`) wraps the same path in a trainable module:
  Gaussian head (the reference's `final` MLP, PyTorch) -> hot path -> head backward -> [DDP gradient all-reduce over
  RCCL when N > 1 (R9)] -> grad clip -> AdamW, so that the one real exchange step of the path is exercised on GPUs.
Workload at N=1: BASELINE.json configs[1] (C2): transformer config, 128 Gaussians per object (1024 pts -> 128
groups), 256x256, batch 32 per GPU x 4 supervised views = 128 rendered views per step.  Weak scaling: every
rank renders its own 32 objects.  Prints ONE JSON line on rank 0.

`--gpus N` with N > 1 and no torch.distributed.run environment starts the N ranks itself (one process per GPU, localhost
rendezvous on a free port) the way the reference's launcher does (pointcept/engines/launch.py:75-88); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks come from the environment.
`value` is the collective-free render-loss hot path at every N (the path shards by object with no exchange step); the step
that CONTAINS the path's one exchange (DDP all-reduce of the 117.9 MB of backbone + head gradients over RCCL, R9) is timed as
`train_step_e2e_standin` at every N, with the same region's single-rank time measured in the same run beside it.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

# OpenMP workers (the CPU-baseline leg runs the oracle on every host core) must sleep, not spin, once their region ends: spinning
# workers starve the Python launch thread of the secondary regions timed afterwards (measured: 1.8 ms instead of 0.18 ms per step)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from unipre3d_amd import _lib, dp, step, synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(kind: str, P: int, R: float, T: int, HW: int) -> float:
    """Per-view algorithmic bytes of each kernel (DESIGN.md 'Algorithmic bytes').
    forward total 168 P + 76 R + 8 T + 24 HW, backward total 308 P + 76 R + 24 HW."""
    return {
        "preprocess_fwd": 168.0 * P,
        "depth_sort": 36.0 * R,               # key/value write 12 + sort read 12 + sort write 12 of the reference algorithm
        "render_fwd": 40.0 * R + 8.0 * T + 24.0 * HW,
        "render_fb": (40.0 * R + 8.0 * T + 24.0 * HW) + (76.0 * R + 24.0 * HW),   # forward share + backward share
        "render_bwd": 76.0 * R + 24.0 * HW,
        "preprocess_bwd": 308.0 * P,
    }[kind]


PROFILE_ROUND = "r06"
PROFILE_FALLBACK_ROUND = "r05"
# measured instruction-class issue costs on gfx950, cycles per wave64 instruction per SIMD (profiles/r01/valu_instruction_classes.txt,
# tools/ub/ops.hip under rocprofv3 --pmc): FMA / MUL / ADD / MOV 2.13, compare / min / select / DPP 4.08, exp / rcp 8.1; a scalar
# instruction costs the SIMD's issue port about 1.7 (profiles/r01/ub_mixed_streams.txt: fma + s_and = 1.8 x an fma alone)
ISSUE_CYCLES = {"fma_class": 2.13, "other_valu": 4.08, "trans": 8.1, "salu": 1.7}
N_SIMD, SHADER_CLOCK_HZ = 1024, 2.4e9     # MI355X: 256 CUs x 4 SIMDs; MI355X_MICROARCH.md peak engine clock


def _profile_json(name: str):
    """Newest committed profile of that name (this round's, else the previous round's: a kernel the round did not touch keeps its
    pass); the round it came from travels with it (`_round`) into every source label."""
    for rnd in (PROFILE_ROUND, PROFILE_FALLBACK_ROUND):
        path = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(path):
            d = json.load(open(path))
            d["_round"] = rnd
            return d
    return None


def pmc_traffic(kernel: str, config: str, default_path: bool):
    """HBM bytes per launch of the dominant tile kernel TOGETHER WITH the bwd_reduce kernel of its scope, from the committed PMC
    passes (profiles/r02/pmc_traffic_<config>.json: separate FETCH_SIZE / WRITE_SIZE runs of this very command, FETCH_SIZE doubled
    per the gfx950 calibration of MI355X_MICROARCH.md).  None when the run does not match a profiled workload."""
    prof = _profile_json(f"pmc_traffic_{config}.json") if default_path else None
    if not prof:
        return None
    per = prof["per_launch"]
    key = {"render_fb": "render_fb_wave_kernel"}.get(kernel)
    if key not in per:
        return None
    reduce_bytes = sum(per[k]["hbm_bytes_corrected"] for k in ("bwd_reduce1_kernel", "bwd_reduce_kernel") if k in per)
    return per[key]["hbm_bytes_corrected"] + reduce_bytes


def issue_roofline(config: str, default_path: bool, measured_ms: float):
    """The bound that actually binds the tile kernel: VALU / SALU ISSUE, not HBM.  From the committed SQ counter passes
    (profiles/r02/sq_issue_<config>.json, tools/profile_r02.sh) the kernel's instruction counts per launch, split into the
    measured issue classes; floor = sum(count x cycles per class) / (1024 SIMDs x clock).  `frac` = floor / measured duration of
    the tile kernel alone (rocprofv3 average of the same profile).  Two floors are given: `lower` prices every non-transcendental
    VALU instruction as an FMA (no kernel with this instruction COUNT can run faster), `by_class` prices add/mul/fma at 2.13,
    transcendentals at 8.1 and the rest (compares, selects, min, DPP-carrying adds are undercounted here) at 4.08."""
    prof = _profile_json(f"sq_issue_{config}.json") if default_path else None
    if not prof:
        return None
    c = prof["per_launch"]
    valu, salu, trans = c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    fma = c.get("SQ_INSTS_VALU_FMA_F32", 0.0) + c.get("SQ_INSTS_VALU_MUL_F32", 0.0) + c.get("SQ_INSTS_VALU_ADD_F32", 0.0)
    per = N_SIMD * SHADER_CLOCK_HZ
    lower = ((valu - trans) * ISSUE_CYCLES["fma_class"] + trans * ISSUE_CYCLES["trans"] + salu * ISSUE_CYCLES["salu"]) / per
    by_class = (fma * ISSUE_CYCLES["fma_class"] + trans * ISSUE_CYCLES["trans"] + max(valu - fma - trans, 0.0) * ISSUE_CYCLES["other_valu"]
                + salu * ISSUE_CYCLES["salu"]) / per
    kernel_ms = prof.get("kernel_avg_ms") or measured_ms
    # shader cycles of the launch as the counters saw them (GRBM_GUI_ACTIVE sums the 8 XCDs); falls back to duration x nominal clock
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0 if c.get("GRBM_GUI_ACTIVE") else kernel_ms * 1e-3 * SHADER_CLOCK_HZ
    lower, by_class = lower * SHADER_CLOCK_HZ / cycles * kernel_ms * 1e-3, by_class * SHADER_CLOCK_HZ / cycles * kernel_ms * 1e-3   # floors in measured cycles -> seconds
    # INDEPENDENT floors: the guide's rate, v_fma_f32 wave64 = 2 cycles per SIMD (MI355X_MICROARCH.md, per-instruction
    # constants), every VALU instruction priced as one -- and the same with transcendentals at quarter rate (8 cycles); no SALU, no
    # builder-calibrated class costs.  In the launch's own measured shader cycles.
    valu_2cyc = valu * 2.0 / N_SIMD / cycles
    valu_tq = ((valu - trans) * 2.0 + trans * 8.0) / N_SIMD / cycles
    # the floor the MEASURED per-opcode issue costs give (tools/ub/opcost.hip -> tools/isa_reconcile.py; C2 only) and the wave-slot occupancy
    # of the launch (profiles/<round>/sq_wait_<config>.json): what separates the kernel from that floor is vacancy, not instruction cost
    extra = {}
    ih = _profile_json("isa_histogram_render_fb.json") if config == "C2" else None
    if ih:
        fl = ih["issue_floor"]["measured_opcode_costs_whole_kernel"]
        extra.update({"frac_opcode_costs": fl["frac"], "floor_us_opcode_costs": fl["frac"] * 1e3 * kernel_ms,
                      "opcode_costs_source": f"profiles/{ih['_round']}/isa_histogram_render_fb.json (static ISA x measured trip counts x tools/ub/opcost.hip costs)"})
    sw = _profile_json(f"sq_wait_{config}.json")
    if sw and sw.get("derived"):
        extra.update({"wave_slot_occupancy": sw["derived"]["wave_slot_occupancy"], "share_of_wave_time_in_s_waitcnt": sw["derived"]["share_of_wave_time_in_s_waitcnt"],
                      "share_of_wave_time_waiting_for_issue": sw["derived"]["share_of_wave_time_waiting_for_issue"]})
    return {**extra, "frac_valu_2cyc": valu_2cyc, "frac_valu_trans_quarter": valu_tq,
            "floor_us_valu_2cyc": 1e3 * valu_2cyc * kernel_ms, "floor_us_valu_trans_quarter": 1e3 * valu_tq * kernel_ms,
            "bound": "valu_issue", "kernel": "render_fb_wave_kernel", "valu_instructions_per_launch": valu, "salu_instructions_per_launch": salu,
            "transcendental_per_launch": trans, "fma_mul_add_per_launch": fma, "floor_ms_lower": 1e3 * lower, "floor_ms_by_class": 1e3 * by_class,
            "kernel_ms_rocprof": kernel_ms, "frac_lower": 1e3 * lower / kernel_ms, "frac_by_class": 1e3 * by_class / kernel_ms,
            "cycles_per_valu_instruction_per_simd": cycles * N_SIMD / valu, "shader_cycles_per_launch": cycles, "class_cycles": ISSUE_CYCLES,
            "reading": ("three floors, none fitted to this kernel: frac_valu_2cyc prices EVERY VALU instruction at the guide's v_fma_f32 rate (2 cycles per wave64 "
                        "instruction and SIMD), frac_valu_trans_quarter prices the transcendentals at a quarter of that rate, frac_opcode_costs uses the per-opcode "
                        "costs measured by tools/ub/opcost.hip (compares / selects / DPP adds ~2.8 cycles, exp2 / rcp ~5.5, a scalar instruction ~1.2 of SIMD time); "
                        "frac_lower / frac_by_class are rounds 2-5's builder-calibrated class costs, kept for continuity"),
            "source": f"profiles/{prof['_round']}/sq_issue_{config}.json (committed rocprofv3 --pmc passes of this command; NOT measured in this "
                      f"run; class costs: builder-calibrated micro-benchmarks, profiles/r01/valu_instruction_classes.txt)"}


def cpu_baseline(batch, H, W, min_seconds=10.0, max_views=None):
    """The CPU restatement (oracle/, kind 'port') timed on a bounded sample of the SAME workload: forward + focal-L2 + backward of
    the batch's views, ONE VIEW PER HOST THREAD (the oracle's inner OpenMP regions run serially inside each: P = 128 Gaussians
    per view is far too little work to fork 128 threads over, which is what rounds 1-2 timed -- 58 views/s of fork/join overhead),
    repeated until >= min_seconds.  The old form (all threads inside each view, views one after another) is reported beside it."""
    import concurrent.futures as cf
    import numpy as np
    from oracle import oracle
    from unipre3d_amd import head, losses
    oracle.build()
    g = synthetic.gaussians_from_batch(batch)
    t = math.tan(batch.fov_deg * math.pi / 360)
    V = batch.world_view.shape[1]
    cores = oracle.num_threads()
    views = [(b, v) for b in range(batch.raw.shape[0]) for v in range(V)]
    if max_views:
        views = views[:max_views]
    args = []
    for (b, v) in views:
        shs = head.concat_sh(g["features_dc"][b], g["features_rest"][b]).numpy()
        args.append(dict(means3D=g["xyz"][b].numpy(), opacities=g["opacity"][b].numpy(), viewmatrix=batch.world_view[b, v].numpy(),
                         projmatrix=batch.full_proj[b, v].numpy(), campos=batch.camera_center[b, v].numpy(), bg=batch.bg.numpy(),
                         image_height=H, image_width=W, tanfovx=t, tanfovy=t, shs=shs, scales=g["scaling"][b].numpy(),
                         rotations=g["rotation"][b].numpy(), sh_degree=1, dtype=np.float32))

    def one(j, inner_threads):
        oracle.set_num_threads(inner_threads)
        a, (b, v) = args[j], views[j]
        r = oracle.forward(**a)
        x = torch.from_numpy(r.color)[None].requires_grad_(True)
        loss = losses.render_loss(x, batch.gt[b, v][None], "focal_l2")
        (gx,) = torch.autograd.grad(loss, x)
        oracle.backward(r, gx[0].numpy())
        r.close()

    torch_threads = torch.get_num_threads()
    torch.set_num_threads(1)                                   # (the per-view loss is a 256 x 256 elementwise op: no intra-op team either)
    workers = max(1, min(cores, len(views)))
    n, t0 = 0, time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=workers) as pool:
        while True:
            list(pool.map(lambda j: one(j, 1), range(len(views))))
            n += len(views)
            el = time.perf_counter() - t0
            if el >= 0.7 * min_seconds:
                break
    torch.set_num_threads(torch_threads)
    # the rounds 1-2 form on 8 views, for continuity
    n2, t2 = 0, time.perf_counter()
    while True:
        for j in range(min(8, len(views))):
            one(j, cores)
            n2 += 1
            el2 = time.perf_counter() - t2
            if el2 >= 0.3 * min_seconds:
                break
        if el2 >= 0.3 * min_seconds:
            break
    return {"value": n / el, "unit": "views/s", "cores": workers, "kind": "port",
            "sample": f"{len(views)} views of the bench batch (fwd + focal_l2 + bwd), one view per host thread ({workers} threads, inner OpenMP "
                      f"regions serial), repeated for {el:.1f} s = {n} renders",
            "all_threads_inside_each_view": {"value": n2 / el2, "unit": "views/s", "cores": cores,
                                             "sample": f"up to {min(8, len(views))} views one after another, {cores} OpenMP threads inside each, {el2:.1f} s = {n2} renders "
                                                       "(the rounds 1-2 form: fork / join dominated at P = 128)"}}


def consumed_roofline(dom, Rc, tiles, HW, NV, avg_ms):
    """The tile kernel's scope priced on the instances the tiles actually CONSUME (R_c) instead of all R = sum of tiles touched."""
    per_inst = {"render_fb": 116.0, "render_fwd": 40.0, "render_bwd": 76.0}[dom]
    per_pix = {"render_fb": 48.0, "render_fwd": 24.0, "render_bwd": 24.0}[dom]
    by_c = (per_inst * Rc + (8.0 * tiles if dom != "render_bwd" else 0.0) + per_pix * HW) * NV
    return per_inst, by_c, by_c / 1e9 / (avg_ms / 1e3)


def other_config_region(name, compact, rank, dev, steps=30, warmup=5, world=1):
    """The hot-only step (the contractual region's definition: activations -> batched render -> loss -> backward -> dL/d head_out,
    seeded by a plain loss.backward()) of ANOTHER BASELINE config, timed in the default driver run so that the scene-level figures
    (BASELINE configs[2..4]; `--compact`: SURVEY 8d's secondary regime) are driver-observed, not builder-printed.  ~1-2 s each."""
    from unipre3d_amd.fused import render_loss_fused
    cfg = synthetic.CONFIGS[name]
    B, P, V, H, W, level = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"], cfg["level"]
    batch = synthetic.make_batch(B, P, V, H, W, level=level, seed=42 + rank, compact=compact).to(dev)
    loss_kind = "focal_l2" if level == "object" else "l2"
    head_out = batch.raw.permute(0, 2, 1).contiguous().requires_grad_(True)

    def hot():
        head_out.grad = None
        loss, _, _ = render_loss_fused(head_out, batch.center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt, batch.bg,
                                       batch.fov_deg, H, W, level=level, offset_scale=batch.offset_scale, loss_kind=loss_kind,
                                       return_images=False)
        loss.backward()
        return loss.detach()

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.3:
        for _ in range(10):
            hot()
        torch.cuda.synchronize()

    def region(profile):
        """`steps` steps between two host barriers; the MAX over ranks of the elapsed time (every rank runs its own B objects: weak scaling,
        no collective inside -- the contractual region's bracket)."""
        for _ in range(warmup):
            hot()
        torch.cuda.synchronize()
        dp.host_barrier()
        if profile:
            _lib.profile_begin(8 * (steps + 2), ("render_fb",), stride=2)
        t0 = time.perf_counter()
        for _ in range(steps):
            l = hot()
        torch.cuda.synchronize()
        dp.host_barrier()
        el = dp.host_all_reduce_max(time.perf_counter() - t0)
        return el, (_lib.profile_end() if profile else None), l

    el, prof, l = region(True)
    reps = sorted(1e3 * region(False)[0] / steps for _ in range(3))
    NV, tiles = B * V, ((W + 15) // 16) * ((H + 15) // 16)
    out = {"ms_per_step": 1e3 * el / steps, "views_s": world * NV * steps / el, "n_gpus": world, "steps": steps, "warmup": warmup, "repeat_min_ms": reps[0],
           "repeat_median_ms": reps[1], "final_loss": float(l),
           "workload": f"{name}{' compact' if compact else ''}: {level}-level, P={P}, {H}x{W}, B={B} x V={V} = {NV} renders/step"
                       + (f" per rank x {world} ranks (max-over-ranks time)" if world > 1 else "") + f", loss {loss_kind}"}
    if rank != 0:
        return out
    ms, cnt = prof["render_fb"]
    if cnt:
        out["tile_kernel_ms"] = ms / cnt
        out["tile_kernel_what"] = "HIP events on the launch stream around render_fb_wave_kernel + its bwd_reduce (every 2nd launch)"
    with torch.no_grad():
        g_used = synthetic.gaussians_from_batch(batch)
        R_mean, walk = _read_num_rendered(g_used, batch, H, W, math.tan(batch.fov_deg * math.pi / 360))
    out["num_rendered_per_view"] = R_mean
    out["walked_mean"] = walk["sorted_positions_walked_per_tile_mean"]
    out["walked_max"] = walk["sorted_positions_walked_per_tile_max"]
    Rc = walk.get("instances_consumed_per_view")
    if cnt and Rc is not None:
        _, by_c, ach = consumed_roofline("render_fb", Rc, tiles, H * W, NV, ms / cnt)
        out["instances_consumed_per_view"] = Rc
        out["frac_consumed"] = ach / HBM_PEAK_GBS
        out["frac_contractual"] = algorithmic_bytes("render_fb", P, R_mean, tiles, H * W) * NV / 1e9 / (ms / cnt / 1e3) / HBM_PEAK_GBS
    return out


def _graph_us(launch, reps, rounds=7):
    """Microseconds per launch of a small operator: `reps` back-to-back launches captured into ONE HIP graph on a side stream and replayed
    between two HIP events recorded on that same stream (the operators launch on torch's current stream, which is the capture / replay
    stream here); minimum and median over `rounds` replays.  A graph keeps the host out of the measurement: the per-call ctypes cost
    (4-6 us) exceeds several of these kernels.  Falls back to events around eager launches if the capture fails."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            launch()
    torch.cuda.synchronize()
    how = "hip_graph"
    try:
        graph = torch.cuda.CUDAGraph()
        # (thread_local: a collective library's watchdog thread querying its events must not invalidate this thread's capture)
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            for _ in range(reps):
                launch()
        run = graph.replay
    except Exception:  # noqa: BLE001
        how = "eager"

        def run():
            for _ in range(reps):
                launch()
    per = []
    with torch.cuda.stream(side):
        run()
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            run()
            e1.record(side)
            e1.synchronize()
            per.append(1e3 * e0.elapsed_time(e1) / reps)
    torch.cuda.synchronize()
    per.sort()
    return per[0], per[len(per) // 2], how


def _cpu_us(fn, min_seconds=0.25, max_calls=50):
    """One-thread wall time of the CPU restatement of the same call (microseconds, best of the repeats)."""
    best, n, t_all = float("inf"), 0, time.perf_counter()
    while n < max_calls and (n == 0 or time.perf_counter() - t_all < min_seconds):
        t0 = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t0)
        n += 1
    return 1e6 * best, n, out


# FPS is a chain of m - 1 DEPENDENT selections; no byte count bounds it.  Floor per selection for ONE workgroup per cloud on gfx950, from the
# latencies of /opt/skills/guides/MI355X_MICROARCH.md (ds_read issue->use ~50 cyc, dependent VALU ~4 cyc, VALU issue 2 cyc per wave64
# instruction): two dependent LDS reads (the winner's key after the barrier, then its coordinates) + one LDS atomic / barrier round (~70) +
# the ~46 dependent instructions of the selection (distance + min + max 10, four DPP steps with their hazard slots 12, readlanes + scalar
# max 8, match + tie key 6, decode + address 10) + the issue time of the other points and waves sharing the SIMD (10 instructions per
# point at 2 cyc).  What the kernel does beyond this model is in DESIGN.md section 7.
def _fps_chain_floor_us(ppt, waves):
    clk = 2 * 50 + 70 + 4 * 46 + max(0, max(1, waves // 4) * ppt - 1) * 10 * 2
    return clk / 2400.0


def pointops_region(dev, seed=11):
    """Row N1 measured: every `pointnet2_batch` operator at the shapes the reference's transformer / pointmlp backbones and
    the ShapeNet loader call it with (openpoints/models/backbone/transformer.py:251-265, layers/group_embed.py:39-57,
    backbone/pointmlp.py:159-163 + decoder three_nn / three_interpolate, dataset/shapenet.py:368), through the C-ABI
    (include/unipre3d_pointops.h) with pre-allocated outputs.  Per operator: microseconds per launch (HIP events around a HIP-graph replay of
    back-to-back launches), algorithmic bytes / time against 8 TB/s -- FPS: microseconds per selection against the dependent-chain floor --,
    the same call on oracle/pointops_oracle.c on ONE host thread (`cpu_baseline`, kind "port"), and whether the two results are bit-equal."""
    import ctypes
    import numpy as np
    from oracle import pointops as opo
    from unipre3d_amd import pointops as po
    lib, ptr = po.load(), _lib.ptr
    g = torch.Generator().manual_seed(seed)

    def cloud(B, N):
        d = torch.randn(B, N, 3, generator=g)
        return (d / d.norm(dim=-1, keepdim=True) * (torch.rand(B, N, 1, generator=g) ** (1 / 3) * 0.5)).contiguous()

    def stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def chk(rc):
        if rc != 0:
            raise RuntimeError(f"pointops C-ABI call failed with code {rc}")

    ops = {}

    def add(name, shape, launch, reps, alg_bytes, cpu_fn, gpu_out, equal, extra=None, atol=None):
        us_min, us_med, how = _graph_us(launch, reps)
        cpu_us, cpu_n, ref = _cpu_us(cpu_fn)
        torch.cuda.synchronize()
        got = [t.cpu().numpy() for t in gpu_out()]
        ref = ref if isinstance(ref, tuple) else (ref,)
        if atol is None:
            same = all(np.array_equal(a.reshape(b.shape), b) for a, b in zip(got, ref))
        else:
            same = all(float(np.abs(a.reshape(b.shape) - b).max()) <= atol * max(1.0, float(np.abs(b).max())) for a, b in zip(got, ref))
        ach = alg_bytes / 1e9 / (us_min * 1e-6)
        e = {"shape": shape, "us": us_min, "us_median": us_med, "timing": f"{how}: {reps} back-to-back launches between two HIP events, min of 7 replays",
             "algorithmic_bytes": alg_bytes,
             "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS},
             "cpu_baseline": {"value": cpu_us, "unit": "us/call", "cores": 1, "kind": "port",
                              "sample": f"the same call on oracle/pointops_oracle.c, one host thread, best of {cpu_n}"},
             "gpu_over_cpu": cpu_us / us_min, "equals_oracle": bool(same), "equality": equal}
        if extra:
            e.update(extra)
        ops[name] = e

    def fps_case(name, B, N, M):
        xyz = cloud(B, N)
        x = xyz.to(dev)
        out = torch.empty(B, M, dtype=torch.int32, device=dev)
        temp = torch.empty(B, N, dtype=torch.float32, device=dev) if N > 8192 else None
        ppt, waves = (1, 4) if N <= 256 else (2, 4) if N <= 512 else (4, 4) if N <= 1024 else (4, 8) if N <= 2048 else ((N + 1023) // 1024, 16)
        add(name, f"{B} clouds x {N} points -> {M} samples", lambda: chk(lib.u3d_furthest_point_sampling(B, N, M, ptr(x), ptr(temp), ptr(out), stream())),
            10, 12.0 * B * N + 4.0 * B * M, lambda: opo.furthest_point_sampling(xyz.numpy(), M), lambda: (out,), "indices bit-exact")
        e = ops[name]
        it = max(M - 1, 1)
        floor = _fps_chain_floor_us(ppt, waves)
        e["roofline"] = {"bound": "dependent_chain", "iterations": it, "achieved": e["us"] / it, "floor": floor, "unit": "us/selection",
                         "frac": floor / (e["us"] / it),
                         "what": "m - 1 selections, each needing the previous one's winner: priced per selection against a serial floor built from the "
                                 "microarchitecture guide's latencies (two dependent LDS reads, one LDS-atomic + barrier round, ~46 dependent instructions, "
                                 f"issue time of {ppt} point(s) per lane on {waves} waves)"}
        e["us_per_selection"] = e["us"] / it
        return x, out

    # ---- transformer tokenizer (32 clouds x 1024 -> 128 groups, r = 0.1, k = 32) ----
    B, N, M, K, C = 32, 1024, 128, 32, 384
    fps_case("fps_1024_to_128", B, N, M)
    xyz = cloud(B, N)
    ctr_idx = torch.from_numpy(opo.furthest_point_sampling(xyz.numpy(), M)).long()
    new_xyz = torch.gather(xyz, 1, ctr_idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    x, nx = xyz.to(dev), new_xyz.to(dev)
    bq = torch.empty(B, M, K, dtype=torch.int32, device=dev)
    add("ball_query", f"{B} x {M} queries over {N} points, r = 0.1, nsample = {K}",
        lambda: chk(lib.u3d_ball_query(B, N, M, ctypes.c_float(0.1), K, ptr(nx), ptr(x), ptr(bq), stream())), 20,
        12.0 * B * (N + M) + 4.0 * B * M * K, lambda: opo.ball_query(0.1, K, xyz.numpy(), new_xyz.numpy()), lambda: (bq,), "indices bit-exact")
    idx_h = opo.ball_query(0.1, K, xyz.numpy(), new_xyz.numpy())
    idx = torch.from_numpy(idx_h).to(dev)
    for tag, c in (("xyz", 3), ("c384", C)):
        pts = torch.randn(B, c, N, generator=g)
        pd = pts.to(dev)
        go = torch.empty(B, c, M, K, device=dev)
        add(f"group_points_{tag}", f"({B},{c},{N}) gathered by ({B},{M},{K})",
            lambda pd=pd, go=go, c=c: chk(lib.u3d_group_points(B, c, N, M, K, ptr(pd), ptr(idx), ptr(go), stream())), 20,
            4.0 * B * c * N + 4.0 * B * M * K + 4.0 * B * c * M * K, lambda pts=pts: opo.group_points(pts.numpy(), idx_h), lambda go=go: (go,), "values bit-exact")
    gout = torch.randn(B, C, M, K, generator=g)
    gd, gp = gout.to(dev), torch.zeros(B, C, N, device=dev)

    def group_grad():
        gp.zero_()
        chk(lib.u3d_group_points_grad(B, C, N, M, K, ptr(gd), ptr(idx), ptr(gp), stream()))
    add("group_points_grad_c384", f"({B},{C},{M},{K}) scattered into ({B},{C},{N}) (incl. the zero-fill)", group_grad, 20,
        4.0 * B * C * M * K + 4.0 * B * M * K + 2 * 4.0 * B * C * N, lambda: opo.group_points_grad(gout.numpy(), idx_h, N), lambda: (gp,),
        "fp32 scatter-add order: <= 1e-5 of the largest value", atol=1e-5)
    feats = torch.randn(B, C, N, generator=g)
    fd, ci = feats.to(dev), ctr_idx.to(torch.int32).to(dev)
    gat = torch.empty(B, C, M, device=dev)
    add("gather_points_c384", f"({B},{C},{N}) gathered by ({B},{M})", lambda: chk(lib.u3d_gather_points(B, C, N, M, ptr(fd), ptr(ci), ptr(gat), stream())), 20,
        4.0 * B * C * M * 2 + 4.0 * B * M, lambda: opo.gather_points(feats.numpy(), ctr_idx.numpy().astype(np.int32)), lambda: (gat,), "values bit-exact")

    # ---- pointmlp (16 clouds x 2048; stages 1024 / 512 / 256 / 128; decoder 2048 <- 512) ----
    B2 = 16
    for n_, m_ in ((2048, 1024), (1024, 512), (512, 256), (256, 128)):
        fps_case(f"fps_{n_}_to_{m_}", B2, n_, m_)
    unk, kn = cloud(B2, 2048), cloud(B2, 512)
    ud, kd = unk.to(dev), kn.to(dev)
    d2, i3 = torch.empty(B2, 2048, 3, device=dev), torch.empty(B2, 2048, 3, dtype=torch.int32, device=dev)
    add("three_nn", f"{B2} x 2048 unknown <- 512 known", lambda: chk(lib.u3d_three_nn(B2, 2048, 512, ptr(ud), ptr(kd), ptr(d2), ptr(i3), stream())), 20,
        12.0 * B2 * (2048 + 512) + 24.0 * B2 * 2048, lambda: opo.three_nn(unk.numpy(), kn.numpy()), lambda: (d2, i3), "squared distances and indices bit-exact")
    d2h, i3h = opo.three_nn(unk.numpy(), kn.numpy())
    rec = 1.0 / (np.sqrt(d2h) + 1e-8)
    wh = (rec / rec.sum(axis=2, keepdims=True)).astype(np.float32)
    C2 = 256
    kf = torch.randn(B2, C2, 512, generator=g)
    kfd, wd, i3d = kf.to(dev), torch.from_numpy(wh).to(dev), torch.from_numpy(i3h).to(dev)
    io = torch.empty(B2, C2, 2048, device=dev)
    add("three_interpolate", f"({B2},{C2},512) -> ({B2},{C2},2048)", lambda: chk(lib.u3d_three_interpolate(B2, C2, 512, 2048, ptr(kfd), ptr(i3d), ptr(wd), ptr(io), stream())),
        20, 4.0 * B2 * C2 * (512 + 2048) + 24.0 * B2 * 2048, lambda: opo.three_interpolate(kf.numpy(), i3h, wh), lambda: (io,), "values bit-exact")
    gi = torch.randn(B2, C2, 2048, generator=g)
    gid, gk = gi.to(dev), torch.zeros(B2, C2, 512, device=dev)

    def interp_grad():
        gk.zero_()
        chk(lib.u3d_three_interpolate_grad(B2, C2, 2048, 512, ptr(gid), ptr(i3d), ptr(wd), ptr(gk), stream()))
    add("three_interpolate_grad", f"({B2},{C2},2048) scattered into ({B2},{C2},512) (incl. the zero-fill)", interp_grad, 20,
        4.0 * B2 * C2 * 2048 + 24.0 * B2 * 2048 + 2 * 4.0 * B2 * C2 * 512, lambda: opo.three_interpolate_grad(gi.numpy(), i3h, wh, 512), lambda: (gk,),
        "fp32 scatter-add order: <= 1e-5 of the largest value", atol=1e-5)

    # ---- the ShapeNet loader's call (dataset/shapenet.py:368): one raw cloud -> 1024 points ----
    fps_case("fps_loader_8192_to_1024", 1, 8192, 1024)
    return {"ops": ops, "contraction": "fma_llvm (default mode; the other two are covered by tests/test_gpu_pointops.py)",
            "all_equal_oracle": all(o["equals_oracle"] for o in ops.values()),
            "what": "row N1: the 9 pointnet2_batch entry points through the C-ABI at the reference's shapes; these operators move 0.4 - 50 MB per "
                    "call, so all but the grouping of 384 channels are launch- / latency-bound, not HBM-bound (frac says by how much)"}


def fusion_region(dev, seed=12):
    """Row N4(a) measured: FeatureFusion's projection -> z-buffer -> gather chain at the transformer config's size (32 objects x 128 group
    centres, the image branch's 384-channel 128 x 128 map: fusion/feat_fusion.py:58-131, model/gaussian_predictor.py:205-227) through the
    C-ABI (include/unipre3d_fusion.h); CPU baseline = oracle/fusion_oracle.py (numpy restatement pinned to golden G7)."""
    import ctypes
    import numpy as np
    from oracle import fusion_oracle as fo
    from unipre3d_amd import cameras as cams
    from unipre3d_amd import fusion as fu
    from unipre3d_amd.standin import object_intrinsics
    lib, ptr = fu.load(), _lib.ptr
    g = torch.Generator().manual_seed(seed)
    B, N, C, H, W = 32, 128, 384, 128, 128
    d = torch.randn(B, N, 3, generator=g)
    center = d / d.norm(dim=-1, keepdim=True) * (torch.rand(B, N, 1, generator=g) ** (1 / 3) * 0.5)
    wv = torch.stack([cams.orbit_cameras(1, cams.OBJECT_CAMERA_DISTANCE, cams.OBJECT_FOV_DEG, cams.OBJECT_ZNEAR, cams.OBJECT_ZFAR, g)[0][0] for _ in range(B)])
    c2w = torch.linalg.inv(wv).contiguous()
    intr = object_intrinsics(cams.OBJECT_FOV_DEG, H)
    fx, fy, cx, cy = (ctypes.c_float(float(v)) for v in (intr[0][0], intr[1][1], intr[0][2], intr[1][2]))
    cam_h = fo.camera_points(center.numpy(), c2w.numpy())
    feat = torch.randn(B, C, H, W, generator=g)
    cam, fd = torch.from_numpy(cam_h).to(dev), feat.to(dev)
    mapped, sel = torch.empty(B, N, C, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev)
    zbuf = torch.empty((lib.u3d_zbuffer_fusion_zbuf_bytes(B, H, W) + 7) // 8, dtype=torch.int64, device=dev)
    strm = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fwd():
        rc = lib.u3d_zbuffer_fusion_forward(B, N, C, H, W, fx, fy, cx, cy, ptr(cam), ptr(fd), ptr(mapped), ptr(sel), ptr(zbuf), strm())
        if rc != 0:
            raise RuntimeError(f"u3d_zbuffer_fusion_forward failed with code {rc}")
    us_f, us_f_med, how = _graph_us(fwd, 20)
    feat_h = feat.numpy()
    cpu_f, n_f, (m_ref, s_ref) = _cpu_us(lambda: fo.mapped_features(cam_h, feat_h, intr[0][0], intr[1][1], intr[0][2], intr[1][2]), max_calls=5)
    torch.cuda.synchronize()
    same_f = bool(np.array_equal(mapped.cpu().numpy(), m_ref) and np.array_equal(sel.cpu().numpy(), s_ref))
    gm = torch.randn(B, N, C, generator=g)
    gmd, gf = gm.to(dev), torch.full((B, C, H, W), float("nan"), device=dev)      # (the kernel must write every element: start from NaN)

    def bwd():
        rc = lib.u3d_zbuffer_fusion_backward(B, N, C, H, W, ptr(gmd), ptr(sel), ptr(zbuf), ptr(gf), strm())
        if rc != 0:
            raise RuntimeError(f"u3d_zbuffer_fusion_backward failed with code {rc}")
    us_b, us_b_med, _ = _graph_us(bwd, 10)
    us_fill, _, _ = _graph_us(lambda: gf.zero_(), 10)           # the yardstick: a plain zero-fill of the same 805 MB (what ABI 1 started with)
    cpu_b, n_b, g_ref = _cpu_us(lambda: fo.mapped_grad(gm.numpy(), s_ref, B, C, H, W), max_calls=3)
    gf.fill_(float("nan")); bwd(); torch.cuda.synchronize()
    gf_h = gf.cpu().numpy()
    err_b = float(np.abs(gf_h - g_ref).max() / max(float(np.abs(g_ref).max()), 1e-30)) if np.isfinite(gf_h).all() else float("inf")
    won = int((s_ref >= 0).sum())
    # algorithmic bytes: points 16 B, the z-buffer's clear + atomic-min + re-read on the pixels hit, the winners' C gathered values
    # (each its own 4-byte element of a channel plane: the map is channel-major) and the (B,N,C) output written once
    alg_f = 16.0 * B * N * 2 + 4.0 * B * H * W + 8.0 * B * N + 4.0 * won * C + 4.0 * B * N * C + 4.0 * B * N
    # backward: the (B,C,H,W) gradient written once + the winner table read + the winners' rows read
    alg_b = 4.0 * B * C * H * W + 8.0 * B * H * W + 4.0 * won * C + 4.0 * B * N
    rf = lambda by, us: {"bound": "hbm", "achieved": by / 1e9 / (us * 1e-6), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / 1e9 / (us * 1e-6) / HBM_PEAK_GBS}
    cb = lambda us, n, what: {"value": us, "unit": "us/call", "cores": 1, "kind": "port", "sample": f"{what} (oracle/fusion_oracle.py, numpy), best of {n}"}
    return {"shape": f"{B} objects x {N} centres, feature map ({C},{H},{W}); {won} of {B * N} points win their pixel",
            "timing": f"{how}: back-to-back launches between two HIP events, min of 7 replays",
            "forward": {"us": us_f, "us_median": us_f_med, "launches": "memset + zbuf_min + gather", "algorithmic_bytes": alg_f, "roofline": rf(alg_f, us_f),
                        "cpu_baseline": cb(cpu_f, n_f, "mapped_features on the same inputs"), "gpu_over_cpu": cpu_f / us_f, "equals_oracle": same_f,
                        "equality": "mapped features and selection bit-exact"},
            "backward": {"us": us_b, "us_median": us_b_med, "us_plain_zero_fill_of_the_same_bytes": us_fill,
                         "launches": "grad_dense (gather form: every element of the (B,C,H,W) gradient written once) + tie_add",
                         "algorithmic_bytes": alg_b, "roofline": rf(alg_b, us_b),
                         "cpu_baseline": cb(cpu_b, n_b, "mapped_grad on the same inputs"), "gpu_over_cpu": cpu_b / us_b,
                         "rel_max_err_vs_oracle": err_b, "equals_oracle": err_b < 1e-6, "equality": "<= 1e-6 (bit-equal unless two points tie for a pixel)"},
            "what": "row N4(a): 3 launches forward, no host synchronisation (the reference: ~20 ops and two .item() / nonzero syncs)"}


def e2e_region(a, batch, dev, world, rank, B, V, H, W, loss_kind, timed):
    """One END-TO-END synthetic pre-training step: stand-in transformer predictor (unipre3d_amd/standin.py: FPS + ball query +
    grouping (N1), tokenizer, 16 blocks, 2D->3D fusion (N4), final MLP; 29.46 M parameters like the reference) -> hot path
    (fused activations + render + loss + backward) -> DDP all-reduce of 117.9 MB of gradients (+ SyncBN) at N > 1 ->
    NaN-check/clip -> AdamW.  The frozen SD-VAE is not part of it (synthetic decoder features)."""
    from unipre3d_amd import cameras as cams
    from unipre3d_amd.fused import backward_unit, render_loss_fused
    from unipre3d_amd.gradcheck import check_and_clip_gradients
    from unipre3d_amd.standin import PointTransformerStandIn, object_intrinsics
    g = torch.Generator().manual_seed(7 + rank)
    d = torch.randn(B, 1024, 3, generator=g)
    pts = (d / d.norm(dim=-1, keepdim=True) * (torch.rand(B, 1024, 1, generator=g) ** (1 / 3) * 0.5)).to(dev)
    img = torch.randn(B, 128, 128, 128, generator=g).to(dev)
    c2w = torch.linalg.inv(batch.world_view[:, 0]).contiguous()            # row-vector view->world of the input view
    intr = object_intrinsics(cams.OBJECT_FOV_DEG, 128)
    torch.manual_seed(43)
    net = PointTransformerStandIn().to(dev)
    nparam = sum(p.numel() for p in net.parameters())

    def make_step(model, optimizer):
        def e2e_step():
            optimizer.zero_grad(set_to_none=True)
            head_out, center = model(pts, img, c2w, intr)
            loss, _, _ = render_loss_fused(head_out, center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt, batch.bg,
                                           batch.fov_deg, H, W, level="object", offset_scale=1.0, loss_kind=loss_kind, return_images=False)
            loss.backward()
            if check_and_clip_gradients(model.parameters(), 1.0):
                optimizer.step()
            return loss.detach()
        return e2e_step

    steps = max(5, min(a.steps, 20))
    solo = None
    if world > 1:
        # the SAME region on one rank's own batch without any exchange (every rank runs it at the same time, nothing is
        # communicated): the single-GPU figure the DDP step below is to be compared with, measured in the same run
        el1, _, _ = timed(make_step(net, torch.optim.AdamW(net.parameters(), lr=1e-4, eps=1e-15, fused=True)), False, steps=steps, warmup=3)
        solo = {"ms_per_step": 1e3 * el1 / steps, "value": B * V * steps / el1, "unit": "views/s",
                "what": "same step, same process, before the DDP wrap: no SyncBN, no all-reduce (max over the ranks running it concurrently)"}
    # every parameter of the stand-in produces a gradient in every step, so DDP need not walk the graph for unused ones
    # (the reference's predictor needs find_unused_parameters=True, which dp.create_ddp_model defaults to)
    on_rccl = world > 1 and dist.get_backend() == dist.Backend.NCCL
    # (SyncBN's all_gather has no gloo implementation for device tensors: the gloo-on-GPU test hook runs plain BN)
    net = dp.create_ddp_model(net, sync_bn=on_rccl or world == 1, find_unused_parameters=False)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, eps=1e-15, fused=True)
    el, _, l = timed(make_step(net, opt), False, steps=steps, warmup=3)
    out = {"value": world * B * V * steps / el, "unit": "views/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
           "parameters": nparam, "gradient_bytes_all_reduced_per_step": 4 * nparam if world > 1 else 0,
           "rccl_ranks": world if (world > 1 and dist.get_backend() == dist.Backend.NCCL) else 0,
           "collective_backend": (str(dist.get_backend()) if world > 1 else None), "final_loss": float(l),
           "what": "1024-pt clouds -> FPS/ball-query/group (HIP) -> tokenizer + 16 transformer blocks (PyTorch-ROCm) -> 2D->3D fusion "
                   "(HIP) -> final MLP -> fused render-loss fwd+bwd (HIP) -> " + ("DDP bucketed all-reduce + SyncBN over RCCL/xGMI -> " if world > 1 else "")
                   + "NaN-check/clip -> AdamW; frozen SD-VAE excluded (synthetic decoder features)"}
    if solo:
        out["n1_same_region"] = solo
        out["speedup_over_n1_same_region"] = out["value"] / solo["value"]
    if rank == 0:
        try:
            out["gradclip_n4b"] = gradclip_region(net, nparam)
        except Exception as e:  # noqa: BLE001
            out["gradclip_n4b"] = {"error": repr(e)[:300]}
    return out


def gradclip_region(net, nparam, reps=20):
    """N4(b) on the stand-in predictor's own gradient set (29.46 M fp32 values in a few hundred tensors, as left by the last
    step): the reference's formulation (per-parameter isnan / isinf scans with their host syncs + clip_grad_norm_,
    train_network.py:368-390), the HIP multi-tensor pass with its single 32-byte host read, and its host-read-free form."""
    from unipre3d_amd import gradcheck
    params = [p for p in net.parameters() if p.grad is not None]
    keep = [p.grad.clone() for p in params]

    def restore():
        for p, g in zip(params, keep):
            p.grad.copy_(g)

    def ref():
        bad = any(torch.isnan(p.grad).any() or torch.isinf(p.grad).any() for p in params)
        if not bad:
            torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        return not bad

    opt = torch.optim.AdamW(params, lr=0.0, fused=True)
    res = {}
    for name, fn in (("reference_formulation", ref), ("hip_one_host_read", lambda: gradcheck.check_and_clip_gradients(params, 1.0)),
                     ("hip_no_host_read", lambda: gradcheck.check_and_clip_deferred(params, opt, 1.0))):
        for _ in range(3):
            restore(); fn()
        tot = 0.0
        for _ in range(reps):
            restore()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            tot += time.perf_counter() - t0
        res[name + "_ms"] = 1e3 * tot / reps
    st = gradcheck.gradient_state(params, 1.0)
    restore()
    res.update({"tensors": len(params), "values": nparam, "total_norm": st["total_norm"], "clipping_live": st["coef"] < 1.0,
                "bytes_read_stats_pass": 4 * nparam,
                "what": "wall time of one check-and-clip on the stand-in's gradients incl. launch + sync (the pass itself reads 118 MB once: "
                        "~15 us at 8 TB/s, so every variant is latency-, not bandwidth-bound)"})
    return res


def per_view_region(batch, B, P, V, H, W, loss_kind, steps=9):
    """The reference's UNCHANGED call pattern through the drop-in module: head activations in torch, then one
    `render_predicted` -> `GaussianRasterizer` call per object and view (train_network.py:418-446 ->
    gaussian_renderer/__init__.py:13-104), torch.stack, torch loss, loss.backward(): B*V operator forwards and B*V operator
    backwards per step.  Host-bound by construction; reported so that the sentence 'train_network.py calls it unchanged'
    has its own number next to the batched / fused entry points."""
    import types
    from unipre3d_amd import head, losses, renderer
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=batch.fov_deg, training_resolution=H), model=types.SimpleNamespace(max_sh_degree=1))
    raw = batch.raw.clone().requires_grad_(True)
    gt = batch.gt.reshape(B * V, 3, H, W)
    host = [0.0]

    def step_fn():
        raw.grad = None
        t0 = time.perf_counter()
        gs = head.process_object_output(raw, batch.center, batch.offset_scale)
        imgs = []
        for i in range(B):
            pc = {k: v[i].contiguous() for k, v in gs.items()}
            for v in range(V):
                imgs.append(renderer.render_predicted(pc, batch.world_view[i, v], batch.full_proj[i, v], batch.camera_center[i, v], batch.bg, cfg)["render"])
        rendered = torch.stack(imgs)
        loss = losses.render_loss(rendered, gt, loss_kind)
        loss.backward()
        host[0] += time.perf_counter() - t0
        return loss.detach()

    def run(n):
        """n steps, each fenced by a synchronise (the route is host-bound: a step's kernels finish with its issue); returns
        (total seconds, host issue seconds, last loss, sorted per-step milliseconds)."""
        for _ in range(2):
            step_fn()
        torch.cuda.synchronize()
        host[0] = 0.0
        per = []
        for _ in range(n):
            t0 = time.perf_counter()
            l = step_fn()
            torch.cuda.synchronize()
            per.append(1e3 * (time.perf_counter() - t0))
        return 1e-3 * sum(per), host[0], l, sorted(per)

    el, host_el, l, per = run(steps)
    out = {"ms_per_step": 1e3 * el / steps, "ms_per_step_median": per[len(per) // 2], "ms_per_step_min": per[0], "steps": steps,
           "value": B * V * steps / el, "unit": "views/s", "host_issue_ms_per_step": 1e3 * host_el / steps,
           "operator_calls_per_step": 2 * B * V, "us_per_forward_backward_pair": 1e6 * el / steps / (B * V), "final_loss": float(l),
           "what": "reference call pattern unchanged: render_predicted per object and view (renderer.render_predicted: ONE binding call per "
                   "view, u3d_render_view_forward/_backward over the C-ABI), torch.stack, torch loss, loss.backward(); every step fenced by a synchronise"}
    # The operator's OWN share of this route, measured directly instead of as a difference of two host-bound loops.  The
    # probe: the 2 x B*V bare binding calls of a step -- `_C().render_view` forward for every (object, view) on pre-sliced leaves (one private
    # leaf set per view: no gradient-accumulation kernels), then ONE autograd pass over the B*V outputs, i.e. the operator's backward nodes as
    # the engine runs them in the real loop -- with no wrapper, slicing, stack or loss around them.
    #   operator_host_ms: host time to ISSUE that sequence (queue empty before, no synchronisation inside);
    #   operator_gpu_ms:  the same sequence captured in a HIP graph and replayed: the GPU time of the operator's own kernels back to back,
    #                     free of the host gaps that dominate the eager route (HIP-event scopes around eager launches measure those gaps).
    try:
        from unipre3d_amd import rasterizer as _rzb
        Cb = _rzb._C()
        with torch.no_grad():
            gs0 = head.process_object_output(batch.raw, batch.center, batch.offset_scale)
        names = ("xyz", "opacity", "scaling", "rotation", "features_dc", "features_rest")
        leaves = [[[gs0[k][i].clone().requires_grad_(True) for k in names] for v in range(V)] for i in range(B)]
        flat_leaves = [t for i in range(B) for v in range(V) for t in leaves[i][v]]
        cam = [[(batch.world_view[i, v].contiguous(), batch.full_proj[i, v].contiguous(), batch.camera_center[i, v].contiguous()) for v in range(V)]
               for i in range(B)]
        gcols = [torch.ones(3, H, W, device=batch.raw.device) for _ in range(B * V)]
        tanfov = math.tan(batch.fov_deg * math.pi / 360)

        def bare_seq():
            cols = []
            for i in range(B):
                for v in range(V):
                    x = leaves[i][v]
                    wv, fp, cc = cam[i][v]
                    cols.append(Cb.render_view(x[0], x[1], x[2], x[3], x[4], x[5], wv, fp, cc, batch.bg, H, W, tanfov, tanfov, 1.0, 1, renderer._FAST_FLAGS)[0])
            return torch.autograd.grad(cols, flat_leaves, gcols)

        def bare_host():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bare_seq()
            dt = time.perf_counter() - t0
            torch.cuda.synchronize()
            return dt
        bare_host()
        hs = sorted(bare_host() for _ in range(5))
        out["operator_host_ms"] = 1e3 * hs[2]
        out["operator_host_ms_min"] = 1e3 * hs[0]
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                bare_seq()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                bare_seq()
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gr.replay()
            torch.cuda.synchronize()
            out["operator_gpu_ms"] = 1e3 * (time.perf_counter() - t0) / 20
            del gr
        out["operator_what"] = ("the 2 x B*V bare binding calls of a step (render_view forward per object and view on private leaves, then one autograd pass "
                                "over the B*V outputs): operator_host_ms = host time to issue them (median of 5), operator_gpu_ms = the same sequence "
                                "replayed from a HIP graph (its kernels back to back, no host gaps)")
    except Exception as e:  # noqa: BLE001
        out["operator_error"] = repr(e)[:300]
        torch.cuda.synchronize()
    # the control: the SAME loop with a NO-OP operator (same inputs / outputs / autograd node, allocations only, no kernel), i.e. what
    # the wrapper around the operator costs by itself (slicing, zeros_like, SH concat, radii > 0, torch.stack, loss, autograd through
    # all of them); the operator's share of the route is the difference
    from unipre3d_amd import rasterizer as _rz

    class _NullOp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *tensors):
            ctx.shapes = [t.shape for t in tensors]
            ctx.dev = tensors[0].device
            return torch.empty(3, H, W, device=ctx.dev), torch.empty(P, dtype=torch.int32, device=ctx.dev)

        @staticmethod
        def backward(ctx, g, _):
            return tuple(torch.empty(s, device=ctx.dev) for s in ctx.shapes)

    prev = _rz.set_operator_override(lambda *tensors: _NullOp.apply(*tensors) + (None,))
    try:
        el0, host0, _, per0 = run(steps)
    finally:
        _rz.set_operator_override(prev)
    # round 3's op-by-op wrapper body (zeros_like + 0, torch.cat, nn.Module per call, radii > 0 around the per-view operator)
    from unipre3d_amd import renderer as _rd
    _rd.FAST_PATH = False
    try:
        el3, _, _, per3 = run(steps)
    finally:
        _rd.FAST_PATH = True
    # the same unchanged double loop captured ONCE into a HIP graph and replayed: the operator holds no host
    # synchronisation, so the B*V forward and B*V backward calls, torch.stack, the loss and autograd's own kernels become one graph launch.
    try:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("skipped at N > 1 (probe of the single-process call pattern; a capture next to a live process group is not worth the risk to the line)")
        graw = batch.raw.clone().requires_grad_(True)

        def graph_body():
            gs = head.process_object_output(graw, batch.center, batch.offset_scale)
            imgs = []
            for i in range(B):
                pc = {k: v[i].contiguous() for k, v in gs.items()}
                for v in range(V):
                    imgs.append(renderer.render_predicted(pc, batch.world_view[i, v], batch.full_proj[i, v], batch.camera_center[i, v], batch.bg, cfg)["render"])
            l = losses.render_loss(torch.stack(imgs), gt, loss_kind)
            l.backward()
            return l
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                graw.grad = None
                l_eager = graph_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_eager = graw.grad.clone()
        graw.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            l_static = graph_body()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        out["graph_replay_ms"] = 1e3 * (time.perf_counter() - t0) / 20
        out["graph_replay_equals_eager"] = bool(torch.equal(l_static.detach(), l_eager.detach()) and torch.equal(graw.grad, g_eager))
        out["graph_replay_what"] = ("the same double loop (2 x B*V operator calls, torch.stack, loss, autograd) captured once in a torch.cuda.CUDAGraph with "
                                    "static inputs and replayed")
        del graph
    except Exception as e:  # noqa: BLE001
        out["graph_replay_ms"] = None
        out["graph_replay_error"] = repr(e)[:300]
        torch.cuda.synchronize()
    out.update({"noop_operator_ms": 1e3 * el0 / steps, "noop_operator_ms_median": per0[len(per0) // 2], "noop_operator_host_issue_ms": 1e3 * host0 / steps,
                "operator_share_ms": 1e3 * (el - el0) / steps, "operator_share_ms_median": per[len(per) // 2] - per0[len(per0) // 2],
                "op_by_op_wrapper_body_ms": 1e3 * el3 / steps, "op_by_op_wrapper_body_ms_median": per3[len(per3) // 2],
                "noop_what": "the same loop with the operator replaced by an autograd node that only allocates its outputs and gradients: "
                             "the wrapper's own launches; operator_share_ms = ms_per_step - noop_operator_ms is a DIFFERENCE OF TWO HOST-BOUND LOOPS "
                             "(noise of either exceeds it: 0.2 - 2.1 ms observed box to box) -- quote operator_gpu_ms / operator_host_ms instead"})
    return out


COMPACT_LINE_MAX = 6000      # bytes: the driver parses the LAST stdout line and gave up on round 5's 29 KB object
FULL_RECORD = os.environ.get("U3D_BENCH_FULL_JSON") or os.path.join(ROOT, "bench_full.json")


def write_full_record(out):
    """Everything the run measured (per-operator dicts, other_configs detail, prose) -> bench_full.json beside the script (or
    $U3D_BENCH_FULL_JSON, /tmp when the tree is read-only); a copy lands in gpurun_out/ when that directory exists so that it travels
    back from the GPU box."""
    path = None
    for cand in (FULL_RECORD, "/tmp/bench_full.json"):
        try:
            with open(cand, "w") as f:
                json.dump(out, f, indent=1)
            path = cand
            break
        except OSError:
            continue
    scratch = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(scratch) and "U3D_BENCH_FULL_JSON" not in os.environ:
        try:
            with open(os.path.join(scratch, "bench_full.json"), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
    return path


def _r(x, nd=4):
    """Round floats for the compact line (significant digits, not decimals: the values span 1e-9 ... 1e9)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{nd}g}")


def compact_line(out, full_path=None):
    """The contract object: contract keys, `roofline`, `cpu_baseline`, the scale keys and a digest of the other regions as `extra_keys`.
    Prose and per-operator detail stay in the full record.  Guaranteed <= COMPACT_LINE_MAX bytes (the digest is dropped piecewise if a
    future key pushes it over)."""
    g = out.get
    cfg, rf, cb = g("config") or {}, g("roofline") or {}, g("cpu_baseline") or {}
    kern = ((g("render_loss_step_ms") or {}).get("kernels") or {})
    line = {k: (g(k) if k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
                else _r(g(k), 7)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                            "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:420], "global_batch": cfg.get("global_batch"), "views_per_step": cfg.get("views_per_step"),
                      "parallelism": cfg.get("parallelism"), "loss": cfg.get("loss"), "autograd_entry": cfg.get("autograd_entry"),
                      "num_rendered_per_view": _r(cfg.get("num_rendered_per_view"), 7),
                      "instances_consumed_per_view": _r((cfg.get("list_consumption") or {}).get("instances_consumed_per_view"), 7),
                      "walked_per_tile_mean": _r((cfg.get("list_consumption") or {}).get("sorted_positions_walked_per_tile_mean"))}
    if rf:
        dom = rf.get("kernel")
        line["roofline"] = {"bound": rf.get("bound"), "kernel": dom, "achieved": _r(rf.get("achieved"), 6), "peak": rf.get("peak"), "unit": rf.get("unit"),
                            "frac": _r(rf.get("frac"), 6), "traffic": _r(rf.get("traffic"), 7),
                            "algorithmic_bytes_per_launch": _r(rf.get("algorithmic_bytes_per_launch"), 8),
                            "avg_launch_us": _r(1e3 * (kern.get(dom) or {}).get("avg_ms", float("nan")), 5),
                            "launches_timed": (kern.get(dom) or {}).get("launches"),
                            "frac_consumed": _r(rf.get("frac_consumed")), "consumed_bytes_per_launch": _r(rf.get("consumed_bytes_per_launch"), 7),
                            "traffic_source": (str(rf.get("traffic_source")).split(" (")[0] if rf.get("traffic_source") else None),
                            "scope": "tile kernel + its bwd_reduce, HIP events on the launch stream" if dom in ("render_fb", "render_bwd")
                                     else "HIP events on the launch stream"}
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 6), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": str(cb.get("sample", ""))[:260]}
    for k in ("rccl_ranks", "collective_backend", "gradient_bytes_all_reduced_per_step", "scale_ok", "scale_verdict"):
        if k in out:
            line[k] = g(k)
    for k in ("train_region_value", "train_region_ms_per_step", "n1_same_region", "speedup_over_n1_same_region"):
        if k in out:
            line[k] = _r(g(k), 6)
    if "scale_note" in out:
        line["scale_note"] = str(g("scale_note"))[:300]
    if "hot_path" in out:
        line["hot_path"] = {k: _r(v, 7) for k, v in g("hot_path").items()}
    if "secondary_regions" in out:
        line["secondary_regions"] = str(g("secondary_regions"))[:120]
    line["full_record"] = os.path.relpath(full_path, ROOT) if full_path and full_path.startswith(ROOT) else full_path

    # ---- digest of the other regions (never `value`) ----
    ex = {}
    try:
        oc, po, fu, pv = g("other_configs") or {}, (g("pointops") or {}).get("ops") or {}, g("fusion") or {}, g("per_view_dropin") or {}
        fr, ri, rep = g("forward_rasterizer") or {}, g("roofline_issue") or {}, g("repeatability") or {}
        ex["kernels_us"] = {k: _r(1e3 * v["avg_ms"]) for k, v in kern.items()}
        ex["repeat_ms"] = [_r(rep.get(k)) for k in ("min", "median", "max")] if rep else None
        ex["host_issue_us"] = _r(g("hot_step_host_issue_us"))
        ex["final_loss"] = _r(g("final_loss"), 7)
        if "roofline_hbm_actual" in out:
            ex["hbm_actual_frac"] = _r(out["roofline_hbm_actual"].get("frac"))
        if ri:
            ex["issue"] = {k: _r(ri.get(k)) for k in ("frac_valu_2cyc", "frac_valu_trans_quarter", "frac_opcode_costs", "wave_slot_occupancy",
                                                       "valu_instructions_per_launch", "salu_instructions_per_launch", "transcendental_per_launch") if k in ri}
        if fr and "error" not in fr:
            ex["forward_rasterizer"] = {"us": _r(1e3 * fr.get("avg_ms", float("nan"))), "frac": _r(fr.get("frac_of_8TBs")), "frac_pmc": _r(fr.get("frac_pmc_bytes")),
                                        "no_invdepth_us": _r(1e3 * (fr.get("without_inverse_depth") or {}).get("avg_ms", float("nan")))}
        if oc:
            ex["other_configs"] = {k: ([_r(v.get("ms_per_step")), _r(1e3 * v["tile_kernel_ms"]) if "tile_kernel_ms" in v else None, _r(v.get("frac_consumed")),
                                        _r(v.get("views_s"))] if isinstance(v, dict) and "error" not in v else "error") for k, v in oc.items()}
            ex["other_configs_cols"] = "ms_per_step,tile_kernel_us,frac_consumed,views_s"
        if po:
            ex["pointops_us"] = {k: _r(v.get("us")) for k, v in po.items()}
            ex["pointops_frac"] = {k: _r((v.get("roofline") or {}).get("frac"), 3) for k, v in po.items()}
            ex["fps_us_per_selection"] = {k: _r(v.get("us_per_selection"), 3) for k, v in po.items() if "us_per_selection" in v}
            ex["pointops_all_equal_oracle"] = (g("pointops") or {}).get("all_equal_oracle")
        elif "pointops" in out:
            ex["pointops_us"] = "error"
        if "forward" in fu:
            ex["fusion"] = {h: {"us": _r((fu.get(h) or {}).get("us")), "frac": _r(((fu.get(h) or {}).get("roofline") or {}).get("frac"), 3),
                                "equals_oracle": (fu.get(h) or {}).get("equals_oracle")} for h in ("forward", "backward")}
        elif fu:
            ex["fusion"] = "error"
        if pv:
            ex["per_view"] = ({k: _r(pv.get(k)) for k in ("ms_per_step", "noop_operator_ms", "graph_replay_ms", "operator_gpu_ms", "operator_host_ms") if k in pv}
                              if "error" not in pv else "error")
        for k, short in (("train_step_with_head", "train_with_head_ms"), ("hot_step_backward_unit", "backward_unit_ms"),
                         ("hot_path_on_round1_workload", "round1_workload_ms")):
            if isinstance(g(k), dict):
                ex[short] = _r(g(k).get("ms_per_step")) if "error" not in g(k) else "error"
        gc = (g("train_step_e2e_standin") or {}).get("gradclip_n4b")
        if isinstance(gc, dict):
            ex["gradclip_ms"] = {k: _r(v) for k, v in gc.items() if isinstance(v, (int, float))}
    except Exception as e:  # noqa: BLE001
        ex["error"] = repr(e)[:160]
    line["extra_keys"] = ex
    # hard cap: drop digest entries (largest first) until the line fits
    while len(json.dumps(line, separators=(",", ":"))) > COMPACT_LINE_MAX and line["extra_keys"]:
        big = max(line["extra_keys"], key=lambda k: len(json.dumps(line["extra_keys"][k])))
        del line["extra_keys"][big]
        line["extra_keys_dropped"] = line.get("extra_keys_dropped", []) + [big]
    return line


def _scale_verdict(world, e2e):
    """One sentence the reader of a SCALE record can take at face value: "pass" iff the step that CONTAINS the exchange ran over RCCL with one
    rank per GPU and reached >= 0.75 N of its own single-rank time in the same run (north_star: >= 6 x at 8 GPUs)."""
    e2e = e2e or {}
    if world == 1:
        return "n/a (N = 1: nothing is exchanged)" if "value" in e2e else "n/a (N = 1; the train region did not run)"
    if "value" not in e2e:
        return "fail: the region with the exchange did not complete"
    if e2e.get("rccl_ranks") != world:
        return f"fail: rccl_ranks {e2e.get('rccl_ranks', 0)} != n_gpus {world} (backend {e2e.get('collective_backend')}): not an RCCL measurement"
    sp = e2e.get("speedup_over_n1_same_region")
    if sp is None:
        return "fail: no single-rank time of the same region"
    return (f"pass: {sp:.2f} x over one rank's same region >= 0.75 x {world}" if sp >= 0.75 * world
            else f"fail: {sp:.2f} x over one rank's same region < 0.75 x {world}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", choices=sorted(synthetic.CONFIGS))
    ap.add_argument("--compact", action="store_true", help="secondary compact-splat regime (SURVEY 8d)")
    ap.add_argument("--two-pass", action="store_true", help="separate forward and backward launch sequences (images kept)")
    ap.add_argument("--unfused", action="store_true", help="torch activations + torch loss around the batched operator")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end stand-in region")
    ap.add_argument("--hot-only", action="store_true", help="primary timed region only (profiling runs: keeps the per-kernel averages clean)")
    ap.add_argument("--value-region", choices=("hot", "train"), default="hot",
                    help="which timed region the contractual `value` / `ms_per_step` quote: 'hot' (default, the contract's definition: the "
                         "collective-free render-loss hot path) or 'train' (the end-to-end stand-in training step, which at N > 1 CONTAINS the "
                         "DDP all-reduce over RCCL -- use it at every N of a scaling series to get the curve of the step with the exchange in it)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the hot-only steps of C3 / C4 / C5 / C2-compact (`other_configs`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the N1 (`pointops`) and N4a (`fusion`) regions")
    ap.add_argument("--next-rows-only", action="store_true", help="ONLY the N1 / N4a regions (profiling runs: rocprofv3 of this prints their kernels alone)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torch.distributed.run environment: start the ranks ourselves, one per GPU (pointcept/engines/launch.py:75-88)
        sys.exit(dp.launch_script(os.path.abspath(__file__), sys.argv[1:], a.gpus,
                                  share_devices=os.environ.get("U3D_BENCH_SHARE_GPU") == "1"))
    rank, local_rank, world = dp.init_from_env(os.environ.get("U3D_BENCH_BACKEND"))   # default: nccl (= RCCL) on GPUs
    if world != a.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {a.gpus}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    dev = torch.device("cuda", torch.cuda.current_device())
    if a.next_rows_only:
        if rank == 0:
            print(json.dumps({"pointops": pointops_region(dev), "fusion": fusion_region(dev)}), flush=True)
        if world > 1:
            dp.host_barrier()
            dp.shutdown()
        return
    cfg = synthetic.CONFIGS[a.config]
    B, P, V, H, W, level = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"], cfg["level"]
    host_batch = synthetic.make_batch(B, P, V, H, W, level=level, seed=42 + rank, compact=a.compact)
    batch = host_batch.to(dev)
    feat_dim = 384 if level == "object" else 64
    torch.manual_seed(42)  # identical initial weights on every rank
    model = dp.GaussianHead(feat_dim, 128 if level == "object" else 32).to(dev)
    # The trainable head of the secondary regions (train_step_with_head): synthetic backbone features, final layer standardised per
    # channel so that its output starts at the N(0,1) statistics of SURVEY 8d (mean 0, std 1 in every one of the 23 channels).
    feats = torch.randn(B, P, feat_dim, generator=torch.Generator().manual_seed(1000 + rank)).to(dev)
    with torch.no_grad():
        raw0 = model(feats)                                     # (B, 23, P)
        # rounds 0-1 timed the hot path on THIS tensor scaled to unit overall std with the bias dropped -- whose channels are not
        # N(0,1): opacity channel mean +1.0 / std 0.8, i.e. more opaque, shallower tiles than SURVEY 8d's draw.  Kept as an extra.
        head_out_r1 = ((raw0 - model.final[2].bias[None, :, None]) / raw0.std()).permute(0, 2, 1).contiguous()
        mu, sd = raw0.mean(dim=(0, 2)), raw0.std(dim=(0, 2))
        model.final[2].weight.div_(sd[:, None])
        model.final[2].bias.copy_((model.final[2].bias - mu) / sd)
    loss_kind = "focal_l2" if level == "object" else "l2"

    from unipre3d_amd.fused import backward_unit, render_loss_fused
    # The hot path starts from the raw head output of SURVEY 8d's PRIMARY regime: o ~ N(0,1) i.i.d. of shape (B, 23, P)
    # (synthetic.make_batch; `--compact`: the scaling channels drawn as -4 + 0.5 N(0,1)) -- the same Gaussians the CPU baseline renders.
    head_out = batch.raw.permute(0, 2, 1).contiguous().requires_grad_(True)      # (B,P,23) point-major, what `final` emits

    def hot_step():
        """R1 -> R7/R2/R4 -> R8 -> R5: loss and dL/d(head output)."""
        head_out.grad = None
        if a.unfused:
            loss, _ = step.render_loss_forward(head_out.permute(0, 2, 1), batch, H, W, 0, loss_kind)
        else:
            loss, _, _ = render_loss_fused(head_out, batch.center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt,
                                           batch.bg, batch.fov_deg, H, W, level=level, offset_scale=batch.offset_scale,
                                           loss_kind=loss_kind, single_pass=not a.two_pass, return_images=False)
        # the standard entry, what the reference's trainer calls (train_network.py:333).  Since ABI 4 the autograd backward runs the
        # chain-rule kernels itself and multiplies by autograd's grad_output inside them (no d_head * g launch); the only launch a
        # plain loss.backward() adds over fused.backward_unit(loss) is autograd's own ones_like fill (extra `hot_step_backward_unit`)
        loss.backward()
        return loss.detach()

    DOMINANT = ("render_fwd", "render_bwd", "render_fb")

    def timed(fn, profile, steps=None, warmup=None, kinds=DOMINANT, prof_stride=1):
        steps = a.steps if steps is None else steps
        for _ in range(a.warmup if warmup is None else warmup):
            fn()
        torch.cuda.synchronize()
        dp.host_barrier()          # (gloo: the collective-free region never depends on RCCL)
        if profile:
            _lib.profile_begin(8 * (steps + 2) * 8, kinds, stride=prof_stride)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        dp.host_barrier()
        t1 = time.perf_counter()
        prof = _lib.profile_end() if profile else None
        return dp.host_all_reduce_max(t1 - t0), prof, out

    # timed region: HIP events only around every 4th launch of the dominant tile kernel(s) (a recorded scope idles the stream ~4-5 us);
    # the small kernels are timed in a short separate pass and merged into the per-kernel table below
    # untimed pre-warm before the contractual W warmup steps: a fresh box pages the libraries in and ramps its clocks during
    # the first second of work (a first-process run measured 20 % slower than its own repeat without this)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 1.0:
        for _ in range(20):
            hot_step()
        torch.cuda.synchronize()
    elapsed, prof, loss = timed(hot_step, True, prof_stride=4)   # every 4th launch of the dominant kernel carries events
    # run-to-run spread of the same K-step region (not `value`: five further repeats without the event scopes)
    reps = sorted(1e3 * timed(hot_step, False, warmup=0)[0] / a.steps for _ in range(5))
    # host cost of issuing one step (the launches of 16 steps against an empty queue, before any synchronisation)
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    for _ in range(16):
        hot_step()
    host_us = 1e6 * (time.perf_counter() - t_h) / 16
    torch.cuda.synchronize()
    _, prof_small, _ = timed(hot_step, True, steps=10, warmup=2, kinds=("preprocess_fwd", "depth_sort", "preprocess_bwd"))
    prof = {k: (prof[k] if prof[k][1] else prof_small[k]) for k in prof}

    # forward rasterizer alone (north_star: ">= 40 % of the HBM roofline in the forward rasterizer"): the operator's forward
    # kernel (colour + inverse depth + backward state written out) on the same Gaussians, HIP events, outside the timed region
    fwd_only, fwd_err, fwd_nodepth = None, None, None
    try:
        from unipre3d_amd import head as _head
        from unipre3d_amd.rasterizer import rasterize_gaussians_batched as _rgb
        with torch.no_grad():
            g_f = synthetic.gaussians_from_batch(synthetic.SyntheticBatch(**dict(batch.__dict__, raw=head_out.detach().permute(0, 2, 1))))
            shs_f = _head.concat_sh(g_f["features_dc"], g_f["features_rest"])
            t_f = math.tan(batch.fov_deg * math.pi / 360)
            fwd = lambda: _rgb(g_f["xyz"], g_f["opacity"], batch.world_view, batch.full_proj, batch.camera_center, batch.bg, H, W, t_f, t_f,
                               shs=shs_f, scales=g_f["scaling"], rotations=g_f["rotation"], sh_degree=1)
            for _ in range(5):
                fwd()
            torch.cuda.synchronize()
            _lib.profile_begin(256, ("render_fwd",))
            for _ in range(20):
                fwd()
            torch.cuda.synchronize()
            fwd_only = _lib.profile_end()["render_fwd"]
            # the same forward as the per-view wrapper asks for it (u3d_render_view_forward: SH through two pointers, NO inverse-depth
            # plane -- the reference drops that output, gaussian_renderer/__init__.py:89), batched over the same views
            import ctypes
            from unipre3d_amd.rasterizer import _Plan
            NVf = B * V
            plan_f = _Plan(B, V, P, H, W, t_f, t_f, 1.0, 1, 4, _lib.FLAG_ANTIALIASING)
            u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
            sc_f = (u8(plan_f.sizes.geom_bytes), u8(plan_f.sizes.binning_bytes), u8(plan_f.sizes.image_bytes))
            col_f, rad_f = torch.empty(NVf, 3, H, W, device=dev), torch.zeros(NVf, P, dtype=torch.int32, device=dev)
            cc = lambda x: x.contiguous()
            args_f = [cc(g_f["xyz"]), cc(g_f["features_dc"]), cc(g_f["features_rest"]), cc(g_f["opacity"]), cc(g_f["scaling"]), cc(g_f["rotation"]),
                      cc(batch.world_view).reshape(NVf, 16), cc(batch.full_proj).reshape(NVf, 16), cc(batch.camera_center).reshape(NVf, 3)]
            strm = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

            def fwd_nd():
                _lib.check(_lib.load().u3d_render_view_forward(ctypes.byref(plan_f.desc), _lib.ptr(batch.bg), *[_lib.ptr(x) for x in args_f], _lib.ptr(col_f),
                                                               _lib.ptr(rad_f), _lib.ptr(None), *[_lib.ptr(x) for x in sc_f], strm), "u3d_render_view_forward")
            for _ in range(5):
                fwd_nd()
            torch.cuda.synchronize()
            _lib.profile_begin(256, ("render_fwd",))
            for _ in range(20):
                fwd_nd()
            torch.cuda.synchronize()
            fwd_nodepth = _lib.profile_end()["render_fwd"]
    except Exception as e:  # noqa: BLE001
        fwd_err = repr(e)[:300]

    # statistics of the workload (outside the timed region): R = num_rendered
    with torch.no_grad():
        g = synthetic.gaussians_from_batch(batch)
        from unipre3d_amd import head
        from unipre3d_amd.rasterizer import rasterize_gaussians_batched
        t = math.tan(batch.fov_deg * math.pi / 360)
        color, radii, _ = rasterize_gaussians_batched(g["xyz"], g["opacity"], batch.world_view, batch.full_proj, batch.camera_center,
                                                      batch.bg, H, W, t, t, shs=head.concat_sh(g["features_dc"], g["features_rest"]),
                                                      scales=g["scaling"], rotations=g["rotation"], sh_degree=1)
    out = None
    if rank == 0:
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        # R from the Gaussian rects: visible Gaussians x tiles touched is accumulated by the kernel; read it back via radii>0
        # (exact value is in the geom scratch; recompute from an extra forward on a fresh plan for reporting)
        from unipre3d_amd.rasterizer import _Plan
        with torch.no_grad():   # R of the Gaussians the timed steps actually rendered
            g_used = synthetic.gaussians_from_batch(synthetic.SyntheticBatch(**dict(batch.__dict__, raw=head_out.detach().permute(0, 2, 1))))
        R_mean, walk_stats = _read_num_rendered(g_used, batch, H, W, t)
        NV = B * V
        kernels = {}
        for k, (ms, cnt) in prof.items():
            if cnt:
                avg_ms = ms / cnt
                by = algorithmic_bytes(k, P, R_mean, tiles, H * W) * NV
                kernels[k] = {"avg_ms": avg_ms, "launches": cnt, "algorithmic_GB_per_launch": by / 1e9,
                              "achieved_GBs": by / 1e9 / (avg_ms / 1e3)}
        hot = {k: v for k, v in kernels.items() if k in ("render_fwd", "render_bwd", "render_fb")}
        dom = max(hot, key=lambda k: hot[k]["avg_ms"]) if hot else None
        fwd_ms = sum(kernels[k]["avg_ms"] for k in ("preprocess_fwd", "depth_sort", "render_fwd") if k in kernels)
        bwd_ms = sum(kernels[k]["avg_ms"] for k in ("render_bwd", "preprocess_bwd") if k in kernels)
        fb_ms = sum(kernels[k]["avg_ms"] for k in kernels)
        fwd_bytes = (168.0 * P + 76.0 * R_mean + 8.0 * tiles + 24.0 * H * W) * NV
        bwd_bytes = (308.0 * P + 76.0 * R_mean + 24.0 * H * W) * NV
        out = {
            "metric": "rendered_views_per_sec", "value": world * NV * a.steps / elapsed, "unit": "views/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.config}: render-loss hot path (activations + render fwd + loss + render bwd), {level}-level, P={P} Gaussians/object, {H}x{W}, "
                                   f"B={B}/GPU x V={V} views = {NV} renders/GPU/step; head output o ~ N(0,1) i.i.d. (SURVEY 8d primary regime)"
                                   + (", scaling channels -4 + 0.5 N(0,1) (compact-splat regime)" if a.compact else ""),
                       "global_batch": B * world, "views_per_step": NV * world, "parallelism": f"dp{world}",
                       "loss": loss_kind, "num_rendered_per_view": R_mean, "list_consumption": walk_stats,
                       "path": "unfused (torch activations/loss)" if a.unfused else "fused head-activation + render + loss (HIP)",
                       "autograd_entry": "loss.backward()"},
            "render_loss_step_ms": {"rasterizer_kernels_total": fb_ms, "kernels": kernels},
            "final_loss": float(loss),
            "repeatability": {"what": f"five further repeats of the same {a.steps}-step timed region, ms per step", "min": reps[0], "median": reps[2],
                              "max": reps[4], "timed_region_ms": 1e3 * elapsed},
            "hot_step_host_issue_us": host_us,
        }
        default_path = not (a.unfused or a.two_pass)
        prof_cfg = a.config + ("_compact" if a.compact else "")
        if dom:
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": kernels[dom]["achieved_GBs"] / HBM_PEAK_GBS,
                               "traffic": pmc_traffic(dom, prof_cfg, default_path),
                               "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_GB_per_launch"] * 1e9,
                               "scope": ("one HIP-event scope per launch of the tile kernel TOGETHER WITH the bwd_reduce kernel that finishes its "
                                         "gradient accumulation (rocprofv3 lists the two separately: their averages add up to this duration)")
                                        if dom in ("render_fb", "render_bwd") else "one HIP-event scope per launch of the kernel"}
            # the same scope priced on the instances the tiles actually CONSUME (R_c) instead of all R = sum of tiles touched: with the
            # reference's activations every splat covers nearly every tile (R ~ P T) while a tile stops after 15-33 entries, so the
            # contractual figure exceeds the peak at C3-C5 for ANY early-terminating implementation; this one cannot
            Rc = walk_stats.get("instances_consumed_per_view")
            if Rc is not None and dom in ("render_fb", "render_fwd", "render_bwd"):
                per_inst, by_c, ach_c = consumed_roofline(dom, Rc, tiles, H * W, NV, kernels[dom]["avg_ms"])
                out["roofline"]["consumed_bytes_per_launch"] = by_c
                out["roofline"]["achieved_consumed"] = ach_c
                out["roofline"]["frac_consumed"] = ach_c / HBM_PEAK_GBS
                out["roofline"]["frac_consumed_what"] = (f"{per_inst:.0f} B x instances consumed ({Rc:.0f} per view) + per-tile / per-pixel terms of SURVEY 8d, "
                                                         "over the same live duration: an upper bound no early-terminating implementation can exceed 1 on")
            _tp = _profile_json(f"pmc_traffic_{prof_cfg}.json")
            out["roofline"]["traffic_source"] = (f"profiles/{_tp['_round']}/pmc_traffic_{prof_cfg}.json (committed rocprofv3 --pmc passes of this command; "
                                                 "NOT measured in this run)") if out["roofline"]["traffic"] else None
            tr = out["roofline"]["traffic"]
            if tr:
                # what this design really moves (it never materialises the per-instance lists the algorithmic figure prices):
                out["roofline_hbm_actual"] = {"bound": "hbm", "achieved": tr / 1e9 / (kernels[dom]["avg_ms"] / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": tr / 1e9 / (kernels[dom]["avg_ms"] / 1e3) / HBM_PEAK_GBS, "bytes_per_launch": tr,
                                              "ratio_to_algorithmic": tr / (kernels[dom]["algorithmic_GB_per_launch"] * 1e9),
                                              "note": "PMC-measured HBM bytes of the same scope over the same live duration: the kernels are not "
                                                      "HBM-bound, see roofline_issue"}
            ri = issue_roofline(prof_cfg, default_path, kernels[dom]["avg_ms"])
            if ri:
                out["roofline_issue"] = ri
            out["roofline_rasterizer_fwd_bwd"] = {"achieved": (fwd_bytes + bwd_bytes) / 1e9 / (fb_ms / 1e3), "peak": HBM_PEAK_GBS,
                                                  "unit": "GB/s", "frac": (fwd_bytes + bwd_bytes) / 1e9 / (fb_ms / 1e3) / HBM_PEAK_GBS,
                                                  "note": "reference-algorithm bytes fwd (168P+76R+8T+24HW) + bwd (308P+76R+24HW) per view "
                                                          "over the sum of all rasterizer kernel times of a step"}
        if dom and "render_fwd" in kernels:
            out["roofline_forward_rasterizer"] = {"achieved": fwd_bytes / 1e9 / (fwd_ms / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": fwd_bytes / 1e9 / (fwd_ms / 1e3) / HBM_PEAK_GBS,
                                                  "note": "reference-algorithm bytes 168P+76R+8T+24HW per view over the sum of forward kernel times"}
            out["roofline_backward_rasterizer"] = {"achieved": bwd_bytes / 1e9 / (bwd_ms / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                   "frac": bwd_bytes / 1e9 / (bwd_ms / 1e3) / HBM_PEAK_GBS}
        if fwd_only and fwd_only[1]:
            by = algorithmic_bytes("render_fwd", P, R_mean, tiles, H * W) * NV
            ms = fwd_only[0] / fwd_only[1]
            out["forward_rasterizer"] = {"kernel": "render_fwd", "avg_ms": ms, "algorithmic_GB_per_launch": by / 1e9,
                                         "achieved_GBs": by / 1e9 / (ms / 1e3), "frac_of_8TBs": by / 1e9 / (ms / 1e3) / HBM_PEAK_GBS,
                                         "what": "operator forward (u3d_rasterize_forward) alone: colour, inverse depth and the backward's state written to HBM"}
            # the same duration priced on the bytes this forward REALLY moves: PMC HBM bytes of the kernel from the
            # committed pass when there is one, else the per-pixel outputs it must write (colour 12 + final_T 4 + limit 4 + inverse depth 4)
            fp = _profile_json(f"pmc_traffic_fwd_{prof_cfg}.json") if default_path else None
            real = None
            if fp:
                real = next((v["hbm_bytes_corrected"] for k, v in fp["per_launch"].items() if "render_fwd" in k and "true" in k), None)
            src_real = f"profiles/{fp['_round'] if fp else PROFILE_ROUND}/pmc_traffic_fwd_{prof_cfg}.json (committed rocprofv3 --pmc passes; NOT measured in this run)"
            if real is None:
                real, src_real = 24.0 * H * W * NV, "analytic: the 24 B per pixel of outputs the kernel writes (no PMC pass committed for this run's shape)"
            out["forward_rasterizer"].update({"real_bytes_per_launch": real, "frac_pmc_bytes": real / 1e9 / (ms / 1e3) / HBM_PEAK_GBS,
                                              "real_bytes_source": src_real,
                                              "reading": "frac_of_8TBs prices SURVEY 8(d)'s algorithmic bytes (40 R + 8 T + 24 HW per view: the contract's "
                                                         "definition, what the >= 40 % target is quoted on); frac_pmc_bytes prices the bytes that really cross "
                                                         "HBM -- this design never materialises the per-instance lists, so it moves ~4 x fewer"})
            if fwd_nodepth and fwd_nodepth[1]:
                ms_n = fwd_nodepth[0] / fwd_nodepth[1]
                real_n = None
                if fp:
                    real_n = next((v["hbm_bytes_corrected"] for k, v in fp["per_launch"].items() if "render_fwd" in k and "false" in k), None)
                out["forward_rasterizer"]["without_inverse_depth"] = {
                    "avg_ms": ms_n, "frac_of_8TBs": by / 1e9 / (ms_n / 1e3) / HBM_PEAK_GBS,
                    "real_bytes_per_launch": real_n if real_n is not None else 20.0 * H * W * NV,
                    "frac_pmc_bytes": (real_n if real_n is not None else 20.0 * H * W * NV) / 1e9 / (ms_n / 1e3) / HBM_PEAK_GBS,
                    "what": "u3d_render_view_forward (what renderer.render_predicted calls): the tile kernel's variant that does not carry the "
                            "inverse-depth plane the reference's wrapper drops; same algorithmic bytes as above (SURVEY 8d prices 24 B per pixel)"}
        elif fwd_err:
            out["forward_rasterizer"] = {"error": fwd_err}
        if not a.no_cpu_baseline:
            # (at N > 1 too: rank 0 times it while the other ranks wait at the host barrier below, so that every line the driver reads is complete)
            out["cpu_baseline"] = cpu_baseline(host_batch, H, W, a.cpu_seconds)
    if world > 1:
        dp.host_barrier()

    # The contractual line is complete at this point.  The secondary regions below use RCCL (DDP) at N > 1; if one of them
    # stalls, every rank's watchdog ends the process after the budget and rank 0 prints the line without them.
    emitted = threading.Lock()

    def emit():
        """ONE compact JSON line (<= 6 KB: the driver's parser) as the LAST line of stdout; the full record goes to bench_full.json."""
        if emitted.acquire(blocking=False) and rank == 0:
            full_path = write_full_record(out)
            print(json.dumps(compact_line(out, full_path), separators=(",", ":")), flush=True)

    def scale_keys(e2e):
        """SCALE-proofing (the driver runs `bench.py --gpus N` with no other flag and reads one line per N): the default `value` is the
        collective-free hot path, which scales ~N x by construction, so everything about the step that CONTAINS the path's one
        exchange sits at the TOP level of the line, and `scale_ok` says whether those figures exist and really ran over RCCL."""
        e2e = e2e or {}
        ok = "value" in e2e and (world == 1 or e2e.get("rccl_ranks") == world)
        keys = {"rccl_ranks": e2e.get("rccl_ranks", 0), "collective_backend": e2e.get("collective_backend"),
                "gradient_bytes_all_reduced_per_step": e2e.get("gradient_bytes_all_reduced_per_step"),
                "train_region_value": e2e.get("value"), "train_region_ms_per_step": e2e.get("ms_per_step"),
                "n1_same_region": (e2e.get("n1_same_region") or {}).get("value") if world > 1 else e2e.get("value"),
                "speedup_over_n1_same_region": e2e.get("speedup_over_n1_same_region") if world > 1 else (1.0 if "value" in e2e else None),
                "scale_ok": bool(ok),
                "scale_verdict": _scale_verdict(world, e2e),
                "scale_note": ("value = collective-free hot path (shards by object, no exchange); train_region_* = end-to-end stand-in step with the DDP "
                               "all-reduce of 117.9 MB + SyncBN over RCCL inside; n1_same_region = that step on one rank's batch before the DDP wrap, "
                               "same run") if ok else
                              ((f"the region with the exchange ran over backend '{e2e.get('collective_backend')}' with rccl_ranks = {e2e.get('rccl_ranks', 0)} "
                                f"!= n_gpus = {world}: not an RCCL measurement (ranks sharing a device over gloo are a plumbing test)")
                               if "value" in e2e else
                               ("the region with the exchange did not complete" + (f": {e2e.get('error')}" if e2e.get("error") else "")
                                + ("; skipped by flag" if (a.no_e2e or a.hot_only or level != "object") else "")))}
        return keys

    def on_budget():
        if rank == 0:
            out["secondary_regions"] = f"aborted after {extras_budget:.0f} s (primary line unaffected)"
            out.update(scale_keys({"error": "aborted by the watchdog"}))
        emit()
        os._exit(0)

    extras_budget = float(os.environ.get("U3D_BENCH_EXTRAS_BUDGET_S", "300"))
    watchdog = threading.Timer(extras_budget, on_budget)
    watchdog.daemon = True
    watchdog.start()
    # ---- secondary regions (never `value`); a failure here must not lose the primary result ----
    extras = {}
    if not a.unfused:
        try:   # transparency: the same step seeded through fused.backward_unit(loss) (rounds 1-2 quoted `value` on this entry)
            def unit_step():
                head_out.grad = None
                l, _, _ = render_loss_fused(head_out, batch.center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt, batch.bg,
                                            batch.fov_deg, H, W, level=level, offset_scale=batch.offset_scale, loss_kind=loss_kind,
                                            single_pass=not a.two_pass, return_images=False)
                backward_unit(l)
                return l.detach()
            t_pre = time.perf_counter()          # (the device idled through the CPU-baseline leg: ramp its clocks again first)
            while time.perf_counter() - t_pre < 0.5:
                for _ in range(20):
                    unit_step()
                torch.cuda.synchronize()
            el_p, _, _ = timed(unit_step, False, steps=min(a.steps, 50), warmup=5)
            el_q, _, _ = timed(hot_step, False, steps=min(a.steps, 50), warmup=5)      # the headline entry again, back to back with it
            extras["hot_step_backward_unit"] = {"ms_per_step": 1e3 * el_p / min(a.steps, 50),
                                                "plain_loss_backward_same_moment_ms_per_step": 1e3 * el_q / min(a.steps, 50),
                                                "what": "identical step seeded by fused.backward_unit(loss) (cached dL/dloss = 1: no ones_like fill) instead "
                                                        "of the headline's plain loss.backward(), and the headline step re-timed right after it"}
        except Exception as e:  # noqa: BLE001
            extras["hot_step_backward_unit"] = {"error": repr(e)[:300]}
    if not a.unfused and not a.compact and not a.hot_only:
        try:   # continuity with rounds 0-1, whose bench line timed the hot path on the head MODEL's (non-standardised) output
            h1 = head_out_r1.detach().requires_grad_(True)

            def r1_step():
                h1.grad = None
                l, _, _ = render_loss_fused(h1, batch.center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt, batch.bg,
                                            batch.fov_deg, H, W, level=level, offset_scale=batch.offset_scale, loss_kind=loss_kind,
                                            single_pass=not a.two_pass, return_images=False)
                backward_unit(l)
                return l.detach()
            el_1, _, _ = timed(r1_step, False, steps=a.steps, warmup=5)
            extras["hot_path_on_round1_workload"] = {
                "ms_per_step": 1e3 * el_1 / a.steps, "value": world * B * V * a.steps / el_1, "unit": "views/s",
                "what": "the same hot path on the tensor rounds 0-1 timed: output of a random head MLP scaled to unit overall std, whose channels "
                        "are not N(0,1) (opacity channel mean +1.0: more opaque splats, tiles saturate after ~11 instead of ~17 sorted entries); "
                        "`value` now follows SURVEY 8d's i.i.d. N(0,1) draw"}
        except Exception as e:  # noqa: BLE001
            extras["hot_path_on_round1_workload"] = {"error": repr(e)[:300]}
    try:
        if a.hot_only:
            raise RuntimeError("skipped (--hot-only)")
        ddp_head = dp.create_ddp_model(model, sync_bn=False)
        opt = torch.optim.AdamW(ddp_head.parameters(), lr=1e-4, eps=1e-15, fused=True)  # train_network.py:156-158 (group lr 1e-4)
        el, _, l = timed(lambda: step.train_step(ddp_head, feats, batch, opt, H, W, 0, loss_kind, fused=not a.unfused), False)
        extras["train_step_with_head"] = {
            "value": world * B * V * a.steps / el, "unit": "views/s", "ms_per_step": 1e3 * el / a.steps, "final_loss": float(l),
            "what": "Gaussian head MLP fwd/bwd + hot path + " + ("DDP all-reduce (RCCL) + " if world > 1 else "")
                    + "NaN-check/clip_grad_norm + AdamW"}
    except Exception as e:  # noqa: BLE001
        extras["train_step_with_head"] = {"error": repr(e)[:300]}
    if level == "object" and not a.hot_only:
        try:
            extras["per_view_dropin"] = per_view_region(batch, B, P, V, H, W, loss_kind)
            extras["per_view_dropin_ms_per_step"] = extras["per_view_dropin"]["ms_per_step"]
        except Exception as e:  # noqa: BLE001
            extras["per_view_dropin"] = {"error": repr(e)[:300]}
    if level == "object" and not a.no_e2e and not a.hot_only:
        try:
            extras["train_step_e2e_standin"] = e2e_region(a, batch, dev, world, rank, B, V, H, W, loss_kind, timed)
        except Exception as e:  # noqa: BLE001
            extras["train_step_e2e_standin"] = {"error": repr(e)[:300]}
    if a.config == "C2" and not (a.compact or a.unfused or a.two_pass or a.hot_only or a.no_other_configs):
        # the other BASELINE configs' hot-only steps in the driver's own run.  N = 1: all of them.  N > 1 (VERDICT r04
        # item 4): the configs BASELINE.json assigns to that GPU count -- C3 (pointmlp shape: 4 GPUs; also reported at 2), C4 / C5 and their
        # "+ fused pixel-Gaussians" forms at 8 -- every rank on its own per-GPU batch, max-over-ranks time, no collective.
        every = (("C3", ("C3", False)), ("C4", ("C4", False)), ("C5", ("C5", False)), ("C2_compact", ("C2", True)),
                 ("C4_fused", ("C4_fused", False)), ("C5_fused", ("C5_fused", False)))
        chosen = every if world == 1 else tuple(e for e in every if e[0] in (("C3",) if world < 8 else ("C3", "C4", "C5", "C4_fused", "C5_fused")))
        oc = {}
        for key, (cname, comp) in chosen:
            try:
                oc[key] = other_config_region(cname, comp, rank, dev, world=world)
            except Exception as e:  # noqa: BLE001
                oc[key] = {"error": repr(e)[:300]}
                if world > 1:
                    break          # (the ranks would no longer meet at the same barriers)
            torch.cuda.empty_cache()
        extras["other_configs"] = oc

    if rank == 0 and not (a.hot_only or a.no_next_rows):
        # rows N1 and N4(a): per-operator throughput, roofline and CPU-oracle baseline; ~3 s together
        for key, fn in (("pointops", pointops_region), ("fusion", fusion_region)):
            try:
                extras[key] = fn(dev)
            except Exception as e:  # noqa: BLE001
                extras[key] = {"error": repr(e)[:300]}

    watchdog.cancel()
    if rank == 0:
        out.update(extras)
        e2e = extras.get("train_step_e2e_standin") or {}
        out.update(scale_keys(e2e))
        if a.value_region == "train" and "value" in e2e:
            # quote the step that contains the exchange; the hot path's own figures move under `hot_path`
            out["hot_path"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "steps": out["steps"], "unit": "views/s"}
            out.update(value=e2e["value"], ms_per_step=e2e["ms_per_step"], steps=e2e["steps"], warmup=3)
            out["config"]["workload"] = ("END-TO-END stand-in training step around the hot path (--value-region train): " + e2e["what"]
                                         + "; per rank: " + out["config"]["workload"])
    emit()
    if world > 1:
        dp.host_barrier()
        dp.shutdown()


def _read_num_rendered(g, batch, H, W, t):
    """(mean over views of R = sum over Gaussians of tiles touched, read from the kernel's own counter; list-consumption statistics)."""
    import ctypes
    from unipre3d_amd import head
    from unipre3d_amd.rasterizer import _Plan
    B, P = g["xyz"].shape[:2]
    V = batch.world_view.shape[1]
    dev = g["xyz"].device
    plan = _Plan(B, V, P, H, W, t, t, 1.0, 1, 4, _lib.FLAG_ANTIALIASING | _lib.FLAG_STATS)
    NV = B * V
    color = torch.empty((NV, 3, H, W), device=dev); radii = torch.zeros((NV, P), dtype=torch.int32, device=dev)
    geom = torch.empty(plan.sizes.geom_bytes, dtype=torch.uint8, device=dev)
    binning = torch.empty(plan.sizes.binning_bytes, dtype=torch.uint8, device=dev)
    image = torch.empty(plan.sizes.image_bytes, dtype=torch.uint8, device=dev)
    p = _lib.ptr
    shs = head.concat_sh(g["features_dc"], g["features_rest"])
    c = lambda x: x.contiguous()
    rc = _lib.load().u3d_rasterize_forward(
        ctypes.byref(plan.desc), p(batch.bg), p(c(g["xyz"])), p(c(shs)), p(None), p(c(g["opacity"])), p(c(g["scaling"])),
        p(c(g["rotation"])), p(None), p(c(batch.world_view).reshape(NV, 16)), p(c(batch.full_proj).reshape(NV, 16)),
        p(c(batch.camera_center).reshape(NV, 3)), p(color), p(None), p(radii), p(geom), p(binning), p(image),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "u3d_rasterize_forward")
    torch.cuda.synchronize()
    off = plan.sizes.num_rendered_offset
    nr = geom[off:off + 4 * NV].view(torch.int32).to(torch.float64)
    # how deep the tiles really walk the view's sorted list (what this design's cost follows, SURVEY 8d "list length consumed"):
    # tile_last = last sorted position that contributed to any pixel of the tile (image scratch: final_T | n_contrib | tile_last)
    al = lambda n: ((n + 255) // 256) * 256
    T = ((W + 15) // 16) * ((H + 15) // 16)
    tl = (image[2 * al(4 * NV * H * W):][: 4 * NV * T].view(torch.int32) & 0x7fffffff).to(torch.float64)
    lim = image[al(4 * NV * H * W):][: 4 * NV * H * W].view(torch.int32).to(torch.int64) & 0xffffffff
    sat = (lim != 0xffffffff)
    # instances CONSUMED, as the reference algorithm would count them ("40 B x instances consumed"): for every tile, the
    # entries of ITS list (sorted positions whose tile rectangle covers the tile) up to the tile's last contributing position.
    # binning scratch: sorted_id u32[NG] | sorted_rect uint2[NG] | ...; uniform batch: view v owns pairs [v P, v P + P)
    NG = NV * P
    K = int(min(P, int(tl.max().item())))
    tiles_x = (W + 15) // 16
    consumed = torch.zeros(NV, dtype=torch.float64, device=dev)
    if K > 0:
        if P > 4096:
            # scene level (u3d_rect_indirect): no sorted copy of the rectangles exists; read them through sorted_id from the per-pair
            # array of the geom scratch: depth f32 | xy float2 | conic_op float4 | rgbd float4 | rect uint2 | ...
            ids = binning[: 4 * NG].view(torch.int32).reshape(NV, P)[:, :K].to(torch.int64)
            o_rect = al(4 * NG) + al(8 * NG) + al(16 * NG) + al(16 * NG)
            rect_all = geom[o_rect:][: 8 * NG].view(torch.int32).reshape(NV, P, 2)
            rects = torch.gather(rect_all, 1, ids[..., None].expand(-1, -1, 2)).to(torch.int64)
        else:
            rects = binning[al(4 * NG):][: 8 * NG].view(torch.int32).reshape(NV, P, 2)[:, :K].to(torch.int64)
        x0, y0, x1, y1 = rects[..., 0] & 0xffff, (rects[..., 0] >> 16) & 0xffff, rects[..., 1] & 0xffff, (rects[..., 1] >> 16) & 0xffff
        tix = torch.arange(T, device=dev)
        tx, ty = (tix % tiles_x)[None, :, None], (tix // tiles_x)[None, :, None]
        pos = torch.arange(1, K + 1, device=dev)[None, None, :]
        tlv = tl.reshape(NV, T, 1).to(torch.int64)
        for v0 in range(0, NV, 16):          # (views in slabs: NV x T x K booleans)
            sl = slice(v0, min(NV, v0 + 16))
            cover = (tx >= x0[sl, None, :]) & (tx < x1[sl, None, :]) & (ty >= y0[sl, None, :]) & (ty < y1[sl, None, :]) & (pos <= tlv[sl])
            consumed[sl] = cover.sum(dim=(1, 2)).to(torch.float64)
    stats = {"sorted_positions_walked_per_tile_mean": float(tl.mean().item()), "sorted_positions_walked_per_tile_max": float(tl.max().item()),
             "pixels_saturated_fraction": float(sat.double().mean().item()),
             "saturation_position_mean": float(lim[sat].double().mean().item()) if bool(sat.any()) else None,
             "instances_consumed_per_view": float(consumed.mean().item()),
             "instances_consumed_what": "sum over tiles of the entries of the tile's own list (rectangle covers the tile) up to its last "
                                        "contributing sorted position: what an early-terminating instance-list implementation reads"}
    return float(nr.mean().item()), stats


if __name__ == "__main__":
    main()
