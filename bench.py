#!/usr/bin/env python
"""Headline benchmark: fitting-iterations/s of the AMASS temporal fit (BASELINE.json configs[1]:
opt_amass_temp.py, one 4 s / 30 fps clip = B 119 frames, smoothness + contact + marker + prior
losses, Adam), one independent sequence per GPU.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE full Adam iteration (forward incl. all 10475 vertices/frame, backward, update).
Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (64->64 3x3 conv, fp32-exact operands on the bf16 matrix cores): algorithmic FLOP /
                  launch duration measured live with HIP events on the engine's stream; ``roofline.hbm`` = the same
                  for the vertex stage (lbs_verts_fwd, HBM-bound side, SURVEY 8(d) "report both")
  cpu_baseline -- the oracle (faithful restatement, two SMPL-X forwards like the reference) timed on
                  this node's host cores on a bounded sample (N=1, rank 0 only).
Before the W warm-up steps the graphs of the run are captured and uploaded and the device is brought to its steady
state: the iteration graph is replayed for ``--ramp-ms`` (measured, tools/overhead_probe.py: the same 20-step call
takes 7.9 ms on an idle device and 7.27 ms after ~200 iterations -- DVFS follows the sustained load of the real kernel
mix; hammering one kernel instead made it worse).  The ramp runs the real iteration on the real buffers, then the
sequence is loaded again (parameters, Adam moments and step counter back to their initial values), so the W warm-up
and K timed steps are exactly the first W + K iterations of the fit; ``ramp_iterations`` is reported in the line.
The one-off host costs of the result path (first ``params72`` / all-gather call load their kernels lazily, 0.5 ms) are
paid during the ramp as well.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

DEFAULT_CONV_VARIANT = int(os.environ.get('LEMO_CONV_VARIANT', '9'))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0      # dense bf16 MFMA (guide: ~2.5 PF; AMD's 5 PF headline is 2:1 sparse)
PEAK_HBM_TBS = 8.0                    # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PMC_FILE = os.path.join('profiles', 'r06_pmc_summary.json')   # rocprofv3 --pmc passes of THIS round's final code


def arithmetic_label(conv_variant):
    """``config.arithmetic`` of the bench line: what the encoder's multiplies run on for a given ``conv_variant`` (include/lemo_hip.h,
    lemo_fit_desc.conv_variant).  Every variant >= 4 is the split-f16 scheme (variants 5-9 differ in which layers share a launch, not in
    arithmetic); tests/test_bench_line.py pins that the string says so (variants 6-9 fell through to the fp32-MFMA string)."""
    split_f16 = ('fp32 values and fp32 accumulation throughout; the encoder\'s MFMA layers multiply each fp32 operand as two '
                 'error-compensated fp16 pieces (split-f16: 3 f16-MFMA products, per-workgroup power-of-two scaling; measured error vs '
                 'float64 at the level of an fp32 convolution)')
    fusion = {4: 'one layer per launch (conv_split_kernels.hip)',
              5: 'consecutive 64->64 layers run as fused pairs (conv_pair_kernels.hip)',
              7: 'fused pairs + encoder head (image, layers 0-1) and tail (their backward) one launch each (conv_head_kernels.hip)',
              8: 'fused pairs + encoder head with layer 2 (enc_head3) + tail (conv_head_kernels.hip)',
              9: 'fused pairs + encoder head with layer 2 (enc_head3) and tail with layer 2\'s backward (enc_tail3), conv_head_kernels.hip',
              10: 'every 64->64 layer ONE Winograd F(2x2, 3x3) launch (fp32 transforms, the 16 position GEMMs split-f16; conv_wino_kernels.hip) '
                  'between the fused head (enc_head3) and tail (enc_tail3)'}
    if conv_variant >= 4:
        return split_f16 + '; ' + fusion.get(conv_variant, 'fused launches (variant %d)' % conv_variant)
    if conv_variant == 3:
        return ('fp32 throughout; the 64->64 encoder layers multiply exact fp32 operands as 3 bf16 pieces '
                'each (6 bf16-MFMA products, fp32 accumulate; error vs float64 <= the fp32-MFMA kernel\'s)')
    return 'fp32 throughout (fp32-input MFMA)'


def build_problem(seq_id, B, device, full_vertices, conv_variant=1):
    from lemo_amd import synthetic
    from lemo_amd.assets import load_assets
    from lemo_amd.fitting import AmassTemporalFitter
    from lemo_amd.vposer import make_vposer_weights
    A = load_assets()
    model = synthetic.make_synthetic_smplx(seed=0)
    vw = make_vposer_weights(2)
    fit = AmassTemporalFitter(model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], B, device,
                              full_vertices=full_vertices, conv_variant=conv_variant)
    seq = synthetic.make_synthetic_sequence(seq_id, B=B)
    # target markers = model markers of the perturbed trajectory, via the product path
    fit.load_sequence(seq['target_params'], np.zeros((B, 67, 3), np.float32), seq['contact_lbl'])
    fit.forward()
    torch.cuda.synchronize(device)
    markers = fit.marker_vertices().detach().cpu().numpy().copy()
    fit.load_sequence(seq['init_params'], markers, seq['contact_lbl'])
    return fit, dict(model=model, vposer_w=vw, enc_w=A['enc_w'], ids=A['ids'], Xmean=A['Xmean'], Xstd=A['Xstd'],
                     seq=seq, markers=markers)


def build_problem_emu(seq_id):
    """--emu: the reduced seeded problem of __graft_entry__.small_problem (B = 14, V = 640) on liblemo_emu.so, sequence `seq_id`"""
    import __graft_entry__ as ge
    from lemo_amd import _hip, synthetic
    from lemo_amd.fitting import AmassTemporalFitter
    lib = _hip.HipLib(_hip.EMU_LIB_PATH, is_emu=True)
    p = ge.small_problem()
    B = p['B']
    fit = AmassTemporalFitter(p['model'], p['vposer_w'], p['enc_w'], p['ids'], p['Xmean'], p['Xstd'], B, torch.device('cpu'), full_vertices=True, lib=lib)
    seq = synthetic.make_synthetic_sequence(3 + seq_id, B=B)
    fit.load_sequence(seq['target_params'], np.zeros((B, len(p['ids']['markers67']), 3), np.float32), seq['contact_lbl'])
    fit.forward()
    markers = fit.marker_vertices().detach().cpu().numpy().copy()
    fit.load_sequence(seq['init_params'], markers, seq['contact_lbl'])
    return fit, dict(seq=seq, markers=markers), B


def conv_launcher(fit, stream):
    """closure that launches the engine's dominant kernel on the engine's own buffers with a scratch output: the 64->64 conv
    (layer 10's shape), or -- conv variant 5 -- the fused forward pair of layers (7, 8) (intermediate into a scratch map too)"""
    if 5 <= fit.conv_variant <= 9:
        return lambda: _conv_pair(fit, 7, False, fit.act[7], fit.dact[0], fit.dact[1], stream)
    return lambda: _conv_layer(fit, 9, False, fit.act[9], fit.dact[1], stream)


def events_ms(stream, launch, reps, precondition=None):
    """average duration of `reps` back-to-back launches (HIP events on `stream`), the smallest of three such averages.  Each
    repetition is preceded by `precondition()` (20 iterations of the real fit): the device's clocks follow the load mix
    (DESIGN 6), and fifty launches of one kernel in a row are not the mix the kernel runs in."""
    best = 1e30
    for _ in range(3):
        with torch.cuda.stream(stream):
            if precondition is not None:
                precondition()
            for _ in range(5):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                launch()
            e1.record(stream)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def time_dominant_kernel(fit, stream, reps=50, use_graph=True):
    """Average duration of the dominant launch (:func:`conv_launcher`) on `stream` and its algorithmic FLOPs, measured with HIP events
    around back-to-back launches on ONE hot input buffer (reported as ``kernel_ms_back_to_back``: not what the kernel
    costs inside the iteration, see :func:`time_conv_chain`)."""
    ms = events_ms(stream, conv_launcher(fit, stream), reps, lambda: fit.step(20, use_graph=use_graph))
    fit.dact[0].zero_(); fit.dact[1].zero_()     # scratch again (border must stay zero; interior rewritten each step)
    return ms, 2.0 * fit.H * fit.W * 64 * 64 * 9 * (2 if 5 <= fit.conv_variant <= 9 else 1)


def _conv_layer(fit, l, bwd, src, dst, stream):
    """launch encoder layer l (act[l] -> act[l+1]) or its backward-data twin on the engine's buffers, the way the engine does"""
    from lemo_amd._hip import ptr
    from lemo_amd.priors import ENC_CHANNELS
    lib, H, W, e = fit.lib, fit.H, fit.W, fit.enc
    ci, co = (ENC_CHANNELS[l + 1], ENC_CHANNELS[l]) if bwd else (ENC_CHANNELS[l], ENC_CHANNELS[l + 1])
    bias, aux, epi = (None, ptr(fit.act[l]), 1) if bwd else (ptr(e.b[l]), None, 0)
    w, w2, w3 = (e.wbwd, e.wbwd2, e.wbwd3) if bwd else (e.w, e.w2, e.w3)
    if fit.conv_variant == 10 and ci == 64 and co == 64:
        pack, winv = e.split_pack(l, bwd, 10)
        rc = lib.conv3x3_wino_f16(ptr(src), ptr(pack), winv, ptr(w[l]), bias, aux, ptr(dst), H, W, epi, None, stream.cuda_stream)
    elif fit.conv_variant >= 4:
        pack, winv = e.split_pack(l, bwd, 4)
        rc = lib.conv3x3_mfma_split_f16(ptr(src), ptr(pack), winv, ptr(w[l]), bias, aux, ptr(dst), H, W, ci, co, epi, stream.cuda_stream)
    elif fit.conv_variant == 3:
        rc = lib.conv3x3_mfma_split(ptr(src), ptr(w3[l]), ptr(w[l]), bias, aux, ptr(dst), H, W, ci, co, epi, stream.cuda_stream)
    elif fit.conv_variant == 2:
        rc = lib.conv3x3_mfma_lds(ptr(src), ptr(w[l]), ptr(w2[l]), bias, aux, ptr(dst), H, W, ci, co, epi, stream.cuda_stream)
    else:
        rc = lib.conv3x3_mfma(ptr(src), ptr(w[l]), bias, aux, ptr(dst), H, W, ci, co, epi, fit.conv_variant, stream.cuda_stream)
    lib.check(rc, 'conv layer')


def _conv_pair(fit, l, bwd, src, mid, dst, stream):
    """launch the fused pair the engine launches (conv variant 5): forward layers (l, l+1): act[l] -> act[l+1] (written) -> act[l+2];
    backward-data layers (l, l-1): d(pre l+1) -> d(pre l-1) with act[l], act[l-1] as epilogue operands"""
    from lemo_amd._hip import ptr
    lib, H, W, e = fit.lib, fit.H, fit.W, fit.enc
    if bwd:
        pa, ia = e.split_pack(l, True, 5)
        pb, ib = e.split_pack(l - 1, True, 5)
        rc = lib.conv3x3_pair_f16(ptr(src), ptr(pa), ia, None, ptr(fit.act[l]), None, ptr(pb), ib, None, ptr(fit.act[l - 1]), ptr(dst), H, W, 1,
                                  None, stream.cuda_stream)
    else:
        pa, ia = e.split_pack(l, False, 5)
        pb, ib = e.split_pack(l + 1, False, 5)
        rc = lib.conv3x3_pair_f16(ptr(src), ptr(pa), ia, ptr(e.b[l]), None, ptr(mid), ptr(pb), ib, ptr(e.b[l + 1]), None, ptr(dst), H, W, 0,
                                  None, stream.cuda_stream)
    lib.check(rc, 'conv pair')


def _capture(fit, stream, body):
    lib = fit.lib
    g = C.c_void_p()
    with torch.cuda.stream(stream):
        lib.check(lib.capture_begin(stream.cuda_stream), 'capture_begin')
        try:
            body()
        finally:
            lib.check(lib.capture_end(stream.cuda_stream, C.byref(g)), 'capture_end')
    return g


def time_conv_chain(fit, stream, use_graph=True, reps=5, with_lbs=False, pairs_only=False):
    """In-iteration duration of the dominant kernel: the iteration's own dependency chain of the fourteen 64->64 launches
    (7 forward layers act[3] -> ... -> act[10], then 7 backward-data layers through the two ping-pong gradient maps with
    the saved activations as epilogue operands) on the engine's own buffers -- every layer reads what the previous launch
    just wrote, like in the fit -- captured once and replayed; HIP events around `reps` repetitions of the chain,
    the smallest of three replays, each preceded by 20 real iterations (clocks of the real kernel mix).  Returns
    ms per launch (total / (14 reps)).  With ``with_lbs`` every repetition opens with the all-vertex lbs_verts_fwd launch
    (the difference of the two totals is that kernel's duration with the caches in the state the chain leaves them in)."""
    from lemo_amd._hip import ptr

    def lbs():
        d, t = fit.data, fit._pose_t
        fit.lib.check(fit.lib.lbs_verts_fwd_xs(C.byref(fit.dev.skin), ptr(t['Xg']), ptr(t['XgS']), fit.Bp, ptr(t['A']), d.nj,
                                               ptr(fit.P['transl']), None, d.V, fit.B, ptr(fit.ws['verts']), ptr(fit.ws['v_posed']),
                                               stream.cuda_stream), 'lbs_verts_fwd')

    def body():
        for _ in range(reps):
            if with_lbs:
                lbs()
            if pairs_only:
                # conv variant 5: the six fused launches of the iteration, each reading what the previous one wrote
                for l in (3, 5, 7):
                    _conv_pair(fit, l, False, fit.act[l], fit.act[l + 1], fit.act[l + 2], stream)
                cur = 0
                for l in (9, 7, 5):
                    _conv_pair(fit, l, True, fit.dact[cur], None, fit.dact[1 - cur], stream)
                    cur = 1 - cur
                continue
            for l in range(3, 10):
                _conv_layer(fit, l, False, fit.act[l], fit.act[l + 1], stream)
            cur = 0
            for l in range(9, 2, -1):
                _conv_layer(fit, l, True, fit.dact[cur], fit.dact[1 - cur], stream)
                cur = 1 - cur
    g = _capture(fit, stream, body)
    best = 1e30
    try:
        for _ in range(3):
            with torch.cuda.stream(stream):
                fit.step(20, use_graph=use_graph)
                fit.lib.check(fit.lib.graph_launch(g, stream.cuda_stream), 'graph_launch')
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                fit.lib.check(fit.lib.graph_launch(g, stream.cuda_stream), 'graph_launch')
                e1.record(stream)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
    finally:
        fit.lib.graph_destroy(g)
    return best / reps          # ms per repetition of the chain (14 conv launches, or the 6 fused pairs [+ 1 lbs launch])


def clock_ramp(fit, stream, ms, use_graph):
    """replay the real iteration for ~ms (the caller restores the initial fit state afterwards).  Returns the number
    of iterations run."""
    n = 0
    if ms <= 0 or stream is None:
        return n
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        while (time.perf_counter() - t0) * 1e3 < ms:
            fit.step(20, use_graph=use_graph)
            stream.synchronize()
            n += 20
    return n


def time_vertex_stage(fit, stream, reps=30, use_graph=True):
    """HBM side of the roofline: lbs_verts_fwd over all V vertices x B frames.  Algorithmic bytes per
    launch: blend directions 3V x 506 x 4 (streamed once) + verts and v_posed written (2 x B x V x 12)."""
    from lemo_amd._hip import ptr
    if not fit.full:
        return None
    lib, d = fit.lib, fit.data
    t = fit._pose_t
    # the launch the engine makes: per-frame features also given pre-split (lemo_pose_ws.XgS)
    args = (C.byref(fit.dev.skin), ptr(t['Xg']), ptr(t['XgS']), fit.Bp, ptr(t['A']), d.nj, ptr(fit.P['transl']), None, d.V, fit.B,
            ptr(fit.ws['verts']), ptr(fit.ws['v_posed']))
    ms = events_ms(stream, lambda: lib.check(lib.lbs_verts_fwd_xs(*args, stream.cuda_stream)), reps, lambda: fit.step(20, use_graph=use_graph))
    nbytes = 3.0 * d.V * 506 * 4 + 2.0 * fit.B * d.V * 12
    flops = 2.0 * 128 * 3 * d.V * 512
    return ms, nbytes, flops


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC passes of this round (tools/gpu_pmc.sh -> PMC_FILE;
    counters cannot be read from inside the process): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports
    half the bytes of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section) -> doubled.  None when the file
    of this round has not been produced yet."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None
    for name, d in json.load(open(path)).items():
        if name.startswith(kernel_prefix) and 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            return (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
    return None


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(prob, B, budget_s=20.0):
    """oracle (kind 'port'): faithful iteration incl. the reference's two SMPL-X forwards."""
    from oracle import lemo_oracle as O
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    physical = int(cores)
    so = O.SmplxOracle(prob['model'])
    vw = {k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}
    ew = {k: torch.from_numpy(v) for k, v in prob['enc_w'].items()}
    fit = O.AmassFitOracle(so, vw, ew, prob['ids'], prob['Xmean'], prob['Xstd'], prob['seq']['init_params'],
                           prob['markers'], prob['seq']['contact_lbl'], faithful=True)
    # pick the intra-op thread count that is fastest on this host (all cores is rarely best for
    # these medium-size ops); the count actually used is what `cores` reports
    best, best_t = None, 1e30
    for nt in sorted({min(cores, c) for c in (8, 16, 32, 64, cores)}):
        torch.set_num_threads(nt)
        fit.step()
        t0 = time.time()
        fit.step(); fit.step()
        t = (time.time() - t0) / 2
        if t < best_t:
            best, best_t = nt, t
    cores = best
    torch.set_num_threads(cores)
    for _ in range(2):
        fit.step()
    n, t0 = 0, time.time()
    while n < 40 and (time.time() - t0 < budget_s or n < 3):
        fit.step()
        n += 1
    dt = time.time() - t0
    # single-forward variant: the same oracle evaluating SMPL-X once per iteration
    fit1 = O.AmassFitOracle(so, vw, ew, prob['ids'], prob['Xmean'], prob['Xstd'], prob['seq']['init_params'],
                            prob['markers'], prob['seq']['contact_lbl'], faithful=False)
    fit1.step()
    n1, t1 = 0, time.time()
    while n1 < 10 and (time.time() - t1 < 4.0 or n1 < 3):
        fit1.step()
        n1 += 1
    single = n1 / (time.time() - t1)
    return dict(value=n / dt, unit='fitting-iterations/s', cores=int(cores), threads=int(cores), physical_cores=physical,
                logical_cpus=os.cpu_count(), cpu_model=cpu_model_name(), kind='port', single_forward_value=single,
                thread_sweep='torch.set_num_threads over {8,16,32,64,all physical}; the fastest is used and reported as cores/threads',
                sample=f'{n} iterations of oracle.AmassFitOracle(faithful=True), B={B}, V=10475, after 2 warm-up; '
                       f'single_forward_value: {n1} iterations with one SMPL-X forward per iteration')


def concurrent_probe(fit0, prob0, B, device, k, steps, conv_variant):
    """k independent clips (sequence ids 0..k-1) fitted side by side on this GPU, one engine + stream each: the aggregate rate
    of the same per-clip iteration, measured the way tools/concurrent_clips.py does (bare step() calls on k streams; the
    engines order themselves), every clip checked bit for bit against its solo run.  NOT the headline value -- BASELINE
    configs[1] is one clip per GPU -- but what a dataset-scale run (thousands of clips per GPU) gets."""
    fits, probs = [fit0], [prob0]
    for i in range(1, k):
        f, p = build_problem(i, B, device, full_vertices=True, conv_variant=conv_variant)
        fits.append(f); probs.append(p)
    streams = [torch.cuda.Stream(device) for _ in fits]
    load = lambda f, p: f.load_sequence(p['seq']['init_params'], p['markers'], p['seq']['contact_lbl'])
    for f, s in zip(fits, streams):
        with torch.cuda.stream(s):
            f.prepare(steps); f.prepare(10)
    solo = []
    for f, p, s in zip(fits, probs, streams):
        load(f, p)
        with torch.cuda.stream(s):
            f.step(10); f.step(steps)
        torch.cuda.synchronize(device)
        solo.append((f.losses(), f.params75().clone()))
    best, same = 0.0, True
    for rep in range(3):
        for f, p in zip(fits, probs):
            load(f, p)
        for f, s in zip(fits, streams):
            with torch.cuda.stream(s):
                f.step(10)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for f, s in zip(fits, streams):
            with torch.cuda.stream(s):
                f.step(steps)
        torch.cuda.synchronize(device)
        best = max(best, k * steps / (time.perf_counter() - t0))
        same = same and all(f.losses() == so[0] and torch.equal(f.params75(), so[1]) for f, so in zip(fits, solo))
    assert all(f.nonfinite_step() == 0 for f in fits)
    return {'clips_per_gpu': k, 'value': best, 'unit': 'fitting-iterations/s (aggregate over the clips)', 'steps': steps,
            'bit_identical_to_solo': bool(same),
            'note': 'same iteration per clip (B=119, V=10475, all vertices forwarded), k engines on k streams; every clip compared '
                    'bit for bit with its solo run in this very call; not the headline config (one clip per GPU)'}


def prox_probe(device, steps=300, stage='S3'):
    """BASELINE configs[3]/[4] on ONE GPU: the native PROX window engine (lemo_prox_*: closure + Adam, captured graphs) at
    B = 100, V = 10475, 256^3 SDF, S2 / S3 weights -- optimizer.step(closure) iterations per second.  Not the headline."""
    import __graft_entry__ as ge
    eng, _ = ge.prox_engine_for(ge.prox_full_problem(stage), device, first_batch_flag=False)
    s = torch.cuda.Stream(device)
    with torch.cuda.stream(s):
        eng.step(100, use_graph=True)
    torch.cuda.synchronize(device)
    best = 0.0
    for _ in range(2):
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            eng.step(steps, use_graph=True)
        torch.cuda.synchronize(device)
        best = max(best, steps / (time.perf_counter() - t0))
    assert eng.nonfinite_step() == 0
    return eng, {'value': best, 'unit': 'PROX fitting-iterations/s (optimizer.step(closure), one window)', 'steps': steps,
                 'workload': f'temp_prox/fitting_temp_slide.py window, PROXD_temp_{stage}.yaml weights: B=100 frames, V=10475, 256^3 synthetic SDF, '
                             '245x115 smoothness image, native engine (24 launches / iteration, graph replay)',
                 'total_loss': eng.loss_dict()['total_loss']}


def perframe_probe(device, clips=64, frames=4, steps=100):
    """BASELINE configs[0] (stage 1, opt_amass_perframe.py): frame fits per second with `clips` clips in lockstep through one
    engine (lemo_amd.fitting.BatchedPerFrameFitter; each clip bit-identical to its solo fit) and for one clip alone."""
    from lemo_amd import synthetic
    from lemo_amd.assets import load_assets
    from lemo_amd.fitting import BatchedPerFrameFitter
    from lemo_amd.vposer import make_vposer_weights
    A = load_assets()
    model, vw = synthetic.make_synthetic_smplx(seed=0), make_vposer_weights(2)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'amass_iter.npz'))
    base = g['markers_rec']
    mk = lambda i: (base[(7 * i) % (119 - frames):(7 * i) % (119 - frames) + frames] * (1.0 + 0.002 * (i % 5))).astype(np.float32)
    betas = synthetic.make_synthetic_sequence(0, B=119)['init_params'][0, 6:16]
    out = {}
    for n in (1, clips):
        bf = BatchedPerFrameFitter(model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], device, batch=n)
        cl, bt = [mk(i) for i in range(n)], [betas] * n
        bf.fit_clips(cl, bt, steps=steps); torch.cuda.synchronize(device)             # captures the graphs
        t0 = time.perf_counter()
        bf.fit_clips(cl, bt, steps=steps); torch.cuda.synchronize(device)
        out[n] = n * frames / (time.perf_counter() - t0)
    return {'value': out[clips], 'unit': 'frame fits/s (100 Adam steps each)', 'clips_in_lockstep': clips, 'one_clip_value': out[1],
            'workload': 'opt_amass_perframe.py stage-1 fit: B=1 objective per frame (marker L1 + three L2 priors), V=10475, frames '
                        'of a clip sequential (warm start), clips batched as rows of one engine; every clip bit-identical to its solo fit'}


def ae_probe(device):
    """per-clip infilling-AE finetune (opt_amass_temp.py:154-214: 60 training steps + eval forward at [1,4,210,135]) in ms"""
    from lemo_amd import synthetic
    from lemo_amd.infill import AE, finetune_and_infill
    w = {k: torch.from_numpy(v).to(device) for k, v in synthetic.make_ae_weights(7).items()}
    ae = AE().to(device)
    ae.load_state_dict(w)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 210, 135, generator=g).to(device)
    mask = (torch.ones(210, 135) > 0).to(device)
    # the caller works on its own (capturable) stream, as a per-clip pipeline worker does: the finetune then runs on that stream
    side = torch.cuda.Stream(device)
    torch.cuda.synchronize(device)
    with torch.cuda.stream(side):
        finetune_and_infill(ae, w, x, mask, steps=60)
        torch.cuda.synchronize(device)
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            finetune_and_infill(ae, w, x, mask, steps=60)
            torch.cuda.synchronize(device)
            best = min(best, (time.perf_counter() - t0) * 1e3)
    from lemo_amd import infill
    from lemo_amd.infill import finetune_and_infill_many
    k = infill.AE_CLIPS                                      # clips carried by every launch of one engine (lemo_ae_desc.clips)
    xs = [torch.randn(1, 4, 210, 135, generator=g).to(device) for _ in range(k)]
    with torch.cuda.stream(side):
        finetune_and_infill_many(ae, w, xs, [mask] * k, steps=60)
        torch.cuda.synchronize(device)
        many = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            finetune_and_infill_many(ae, w, xs, [mask] * k, steps=60)
            torch.cuda.synchronize(device)
            many = min(many, (time.perf_counter() - t0) * 1e3 / k)
    # the round-2 path (autograd function + flat Adam under a captured graph), same clip: what the step engine replaced
    finetune_and_infill(ae, w, x, mask, steps=60, engine=False)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    finetune_and_infill(ae, w, x, mask, steps=60, engine=False)
    torch.cuda.synchronize(device)
    old = (time.perf_counter() - t0) * 1e3
    return {'value': best, 'unit': 'ms per clip (60 finetune steps + eval forward)', 'higher_is_better': False,
            'clips_side_by_side': k, 'side_by_side_ms_per_clip': many,
            'path': 'native step engine (lemo_ae_*), 53 launches per step; side by side = k clips carried by every launch of ONE engine on one stream',
            'autograd_path_ms': old,
            'workload': 'models/AE.py infilling autoencoder, [1,4,210,135] clip image, masked L1, Adam 3e-6 (opt_amass_temp.py:154-214)'}


def timed_fit(fit, prob, stream, device, steps=100, warmup=10):
    """iterations/s of a fresh fit over `steps` timed steps after `warmup` (graphs prepared before, result read inside the window)"""
    fit.load_sequence(prob['seq']['init_params'], prob['markers'], prob['seq']['contact_lbl'])
    with torch.cuda.stream(stream):
        fit.prepare(steps); fit.prepare(warmup)
        fit.step(warmup, use_graph=True)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    with torch.cuda.stream(stream):
        fit.step(steps, use_graph=True)
        stream.synchronize()
    _ = fit.params72()
    torch.cuda.synchronize(device)
    return steps / (time.perf_counter() - t1)


def variant_probe(prob_seq_id, B, device, stream, variant, ramp_fit):
    """the same clip on another kernel family of the encoder, same process, same box, clocks brought up by the headline engine first:
    an A/B the line carries itself"""
    f, p = build_problem(prob_seq_id, B, device, full_vertices=True, conv_variant=variant)
    with torch.cuda.stream(stream):
        ramp_fit.step(100, use_graph=True)
    torch.cuda.synchronize(device)
    best = max(timed_fit(f, p, stream, device) for _ in range(2))
    assert f.nonfinite_step() == 0
    return best


def mpjpe_probe(fit, prob, stream, device, steps=100, budget_threads=16):
    """BASELINE's 'MPJPE vs ref' (SURVEY 8(d)): the full 100-step fit on the GPU and on the CPU oracle (one SMPL-X forward per
    iteration -- same arithmetic as the faithful two-forward loop, pinned equal in tests/test_oracle.py -- with the loop's own lr
    switch) from identical inputs; mean over frames and the first 22 joints of ||J_gpu - J_oracle||_2 in mm."""
    from oracle import lemo_oracle as O
    fit.load_sequence(prob['seq']['init_params'], prob['markers'], prob['seq']['contact_lbl'])
    with torch.cuda.stream(stream):
        fit.step(steps, use_graph=True)
        fit.forward()
    torch.cuda.synchronize(device)
    j_gpu = fit.posed_joints().cpu()
    total_gpu = fit.losses()['total']
    torch.set_num_threads(budget_threads)
    so = O.SmplxOracle(prob['model'])
    vw = {k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}
    ew = {k: torch.from_numpy(v) for k, v in prob['enc_w'].items()}
    ofit = O.AmassFitOracle(so, vw, ew, prob['ids'], prob['Xmean'], prob['Xstd'], prob['seq']['init_params'], prob['markers'],
                            prob['seq']['contact_lbl'], faithful=False)
    t0 = time.time()
    first = ofit.step()
    for _ in range(steps - 1):
        last = ofit.step()
    with torch.no_grad():
        _, j_ref, _ = ofit._body(O.convert_to_3D_rot(ofit.params75()))
    return {'value': O.mpjpe_mm(j_gpu, j_ref[:, :55]), 'unit': 'mm (mean over frames and the first 22 joints, GPU fit vs CPU oracle fit)',
            'steps': steps, 'total_loss_start': first['total'], 'total_loss_oracle': last['total'], 'total_loss_gpu': total_gpu,
            'oracle_seconds': time.time() - t0,
            'note': 'free-running 100-step trajectories of a kinked objective under Adam: the per-step parity statements are the '
                    'teacher-forced tests (tests/test_gpu_teacher.py); this is the metric BASELINE.json names'}


def main_prox(args, world, rank, device):
    """--workload prox: BASELINE configs[4]'s per-GPU leg.  Recordings shard over ranks (windows of one recording are
    sequential: temp_prox/main_slide.py:257); every rank fits the current window of ITS recording -- B = 100, V = 10475,
    256^3 SDF, S3 -- for K timed iterations; one all-gather of the fitted per-frame rows (lemo_amd.sharding)."""
    import __graft_entry__ as ge
    from lemo_amd.prox import ENGINE_PARAMS
    from lemo_amd.sharding import gather_fitted_params
    import contextlib
    gpu = not args.emu
    lib = None
    if gpu:
        prob = ge.prox_full_problem('S3')
    else:                                                      # --emu: the reduced seeded window on the host-emulated library
        from lemo_amd import _hip
        lib = _hip.HipLib(_hip.EMU_LIB_PATH, is_emu=True)
        prob = ge.prox_small_problem(stage='S3')
    rng = np.random.default_rng(100 + rank)                    # rank r's own recording: perturbed initial fit and keypoints
    prob['params'] = {k: (np.asarray(v, np.float32) + (rng.standard_normal(np.shape(v)).astype(np.float32) * 0.01 if k != 'betas' else 0))
                      for k, v in prob['params'].items()}
    eng, _ = ge.prox_engine_for(prob, device, first_batch_flag=False, lib=lib)
    s = torch.cuda.Stream(device) if gpu else None
    on_stream = (lambda: torch.cuda.stream(s)) if gpu else contextlib.nullcontext
    sync = (lambda: torch.cuda.synchronize(device)) if gpu else (lambda: None)
    rows = lambda: torch.cat([eng.P[k] for k, _ in ENGINE_PARAMS], dim=1)
    n_warm = max(args.warmup, 100) if gpu else args.warmup

    def barrier():
        if world > 1:
            dist.barrier()
    with on_stream():
        eng.step(n_warm, use_graph=gpu)                          # graphs captured + clocks up
    sync()
    gather_fitted_params(rows()[None])
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    with on_stream():
        eng.step(args.steps, use_graph=gpu)
        if gpu:
            s.synchronize()
    local = rows()
    gathered = gather_fitted_params(local[None])
    sync()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    per_rank = None
    if world > 1:
        allt = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(allt, tmax)
        per_rank = [args.steps / float(t.item()) for t in allt]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert gathered.shape[0] == world and torch.equal(gathered[rank], local) and bool(torch.isfinite(gathered).all())
    for r in range(world):
        if r != rank:
            assert not torch.equal(gathered[r], local), 'ranks fitted the same recording'
    assert eng.nonfinite_step() == 0
    out = {'metric': 'PROX fitting-iterations/sec (100-frame window, PROXD_temp_S3)', 'value': world * args.steps / dt,
           'unit': 'fitting-iterations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': n_warm,
           'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
           'data': 'synthetic',
           'config': {'workload': 'temp_prox/main_slide.py PROXD_temp_S3.yaml: one 100-frame sliding window per GPU (recordings shard '
                                  'over ranks, windows of a recording are sequential), V=10475, 256^3 synthetic SDF, smoothness + '
                                  'infilling priors, native PROX engine', 'frames': 100, 'recordings': world,
                      'parallelism': f'recording-shard x{world} + 1 all_gather'},
           'total_loss': eng.loss_dict()['total_loss']}
    if per_rank is not None:
        out['per_rank_iterations_per_s'] = per_rank
    if not gpu:
        out['config'].update(workload='DRY RUN (--emu): reduced PROX window on the host-emulated kernel library -- control flow only, the '
                                      'numbers mean nothing', frames=int(eng.B), backend=args.backend, emulated=True)
    if rank == 0:
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--frames', type=int, default=119, help='B = clip_seconds*30-1 (the "T=120" clip)')
    ap.add_argument('--active-vertices-only', action='store_true',
                    help='forward only the 253 vertices the losses read (NOT the headline config)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--model', choices=('iid', 'coherent'), default='iid',
                    help='synthetic SMPL-X-shaped model: skinning joints i.i.d. per vertex (the worst case, rounds 1-3) or with the index '
                         'locality of the licensed model (lemo_amd.synthetic._coherent_skinning)')
    ap.add_argument('--conv-variant', type=int, default=DEFAULT_CONV_VARIANT)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ramp-ms', type=float, default=250.0, help='untimed replay of the iteration before the warm-up steps (0 = off)')
    ap.add_argument('--workload', choices=('amass', 'prox'), default='amass',
                    help="amass (default, the headline: BASELINE configs[1]/[2]) or prox (configs[4]'s per-GPU leg: one S3 window per GPU)")
    ap.add_argument('--no-extras', action='store_true', help='skip the non-headline objects (prox_window, perframe, ae_finetune)')
    ap.add_argument('--backend', choices=('nccl', 'gloo'), default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only with --emu)')
    ap.add_argument('--emu', action='store_true',
                    help='DRY RUN of the multi-rank control flow without GPUs (tests/test_sharding.py): a reduced problem on the host-emulated '
                         'kernel library (liblemo_emu.so, CPU tensors), eager launches, no roofline / extras / CPU baseline -- the numbers '
                         'mean nothing; the barriers, the max-over-ranks timing, the one all-gather and the shard self-checks are the real ones')
    ap.add_argument('--concurrent-clips', type=int, default=3,
                    help='after the headline measurement (one clip per GPU), also time this many independent clips fitted side by '
                         'side on GPU 0 (reported as "concurrent_clips", never as "value"; 0 = off)')
    args = ap.parse_args()
    if args.model == 'coherent':
        from lemo_amd import synthetic as _syn
        _syn.DEFAULT_COHERENT = True

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain ``python bench.py --gpus N`` (the way a driver runs --gpus 1; the reference itself is one process looping over clips,
        # opt_amass_temp.py:251): become the launcher -- one rank per GPU under torch.distributed.run, same argv, and pass its exit
        # code on.  The ranks print the ONE JSON line (rank 0) to this process's stdout.
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '2' if args.emu else '8')
        raise SystemExit(subprocess.call(cmd, env=env))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.emu:
        assert args.backend == 'gloo' or world == 1, '--emu runs on CPU tensors: use --backend gloo'
        device = torch.device('cpu')
        args.no_graph, args.no_cpu_baseline, args.no_extras, args.concurrent_clips, args.ramp_ms = True, True, True, 0, 0.0
        torch.set_num_threads(2)
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X (no CPU fallback; --emu is a control-flow dry run, not a fallback)')
        assert args.backend == 'nccl', 'GPU runs use RCCL (backend nccl)'
        device = torch.device('cuda', local_rank)
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.emu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.workload == 'prox':
        main_prox(args, world, rank, device)
        if world > 1:
            dist.destroy_process_group()
        return

    from lemo_amd.sharding import gather_fitted_params
    import contextlib
    gpu = not args.emu
    sync = (lambda: torch.cuda.synchronize(device)) if gpu else (lambda: None)
    if gpu:
        B = args.frames
        fit, prob = build_problem(rank, B, device, full_vertices=not args.active_vertices_only, conv_variant=args.conv_variant)
        stream = torch.cuda.Stream(device)
        on_stream = lambda: torch.cuda.stream(stream)
    else:
        fit, prob, B = build_problem_emu(rank)
        stream, on_stream = None, contextlib.nullcontext
    use_graph = not args.no_graph

    def barrier():
        if world > 1:
            dist.barrier()

    with on_stream():
        if use_graph:                       # record + upload the graphs of all calls (nothing runs): capture is not a step
            fit.prepare(args.warmup)
            fit.prepare(args.steps)
            fit.prepare(20)
    ramp_iters = clock_ramp(fit, stream, args.ramp_ms, use_graph)
    gather_fitted_params(fit.params72()[None])           # result path once, untimed: its torch kernels load lazily
    sync()
    fit.load_sequence(prob['seq']['init_params'], prob['markers'], prob['seq']['contact_lbl'])   # back to iteration 0
    if gpu:
        stream.wait_stream(torch.cuda.current_stream(device))     # (the engine orders its launches behind load_sequence itself too)
    with on_stream():
        fit.step(args.warmup, use_graph=use_graph)
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    with on_stream():
        fit.step(args.steps, use_graph=use_graph)
        if gpu:
            stream.synchronize()
    local72 = fit.params72()
    gathered = gather_fitted_params(local72[None])                 # the path's one collective
    sync()
    barrier()
    dt_local = time.perf_counter() - t0
    tmax = torch.tensor([dt_local], dtype=torch.float64, device=device)
    per_rank = None
    if world > 1:
        allt = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(allt, tmax)
        per_rank = [args.steps / float(t.item()) for t in allt]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # self-check of the sharded run: every rank fitted ITS OWN sequence (rows differ between ranks) and the gathered
    # block of this rank is bit-identical to what it contributed
    assert gathered.shape == (world, B, 72)
    assert torch.equal(gathered[rank], local72), 'all_gather returned a different block for this rank'
    for r in range(world):
        if r != rank:
            assert not torch.equal(gathered[r], local72), f'rank {r} returned the same fit as rank {rank}: sequences not sharded'
    assert bool(torch.isfinite(gathered).all()) and fit.nonfinite_step() == 0
    losses = fit.losses()
    if args.emu:
        out = {'metric': 'fitting-iterations/sec (T=120 frames)', 'value': world * args.steps / dt, 'unit': 'fitting-iterations/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'DRY RUN (--emu): reduced problem on the host-emulated kernel library -- control flow only, the '
                                      'numbers mean nothing', 'frames': B, 'sequences': world, 'parallelism': f'seq-shard x{world} + 1 all_gather',
                          'backend': args.backend, 'emulated': True},
               'final_total_loss': losses['total'], 'per_rank_iterations_per_s': per_rank,
               'shard_self_checks': 'own block bit-identical, other ranks\' blocks differ, all finite'}
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # a second, longer timed window of the same fit (world 1 only): the 20-step default window is 7 ms and the pool's boxes
    # differ by +-5 %; 100 steps is the reference's whole per-clip fit (opt_amass_temp.py:349)
    value_100 = None
    if world == 1 and use_graph:
        fit.load_sequence(prob['seq']['init_params'], prob['markers'], prob['seq']['contact_lbl'])
        with torch.cuda.stream(stream):
            fit.prepare(100)
            fit.step(args.warmup, use_graph=True)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        with torch.cuda.stream(stream):
            fit.step(100, use_graph=True)
            stream.synchronize()
        _ = fit.params72()
        torch.cuda.synchronize(device)
        value_100 = 100.0 / (time.perf_counter() - t1)

    stage_us = None
    if world == 1 and use_graph and gpu:
        with torch.cuda.stream(stream):
            fit.step(40, use_graph=True)                    # clocks of the real kernel mix
            stage_us = fit.stage_census(20)
        torch.cuda.synchronize(device)
    b2b_ms, kern_flops = time_dominant_kernel(fit, stream, use_graph=use_graph)
    pairs = 5 <= fit.conv_variant <= 9
    wino = fit.conv_variant == 10
    if pairs:
        # conv variant 5: the dominant kernel is the fused PAIR (two 64 -> 64 layers per launch, 6 launches per iteration); its
        # algorithmic work is two layers' (the halo recompute of the intermediate tile is overhead, not work)
        chain_ms = time_conv_chain(fit, stream, use_graph=use_graph, pairs_only=True)     # 6 launches per repetition
        kern_ms, n_chain = chain_ms / 6.0, 6
    else:
        chain_ms = time_conv_chain(fit, stream, use_graph=use_graph)                      # 14 launches per repetition
        kern_ms, n_chain = chain_ms / 14.0, 14
    achieved = kern_flops / (kern_ms * 1e-3) / 1e12
    vs = time_vertex_stage(fit, stream, use_graph=use_graph)
    lbs_in_chain_ms = (time_conv_chain(fit, stream, use_graph=use_graph, with_lbs=True, pairs_only=pairs) - chain_ms) if fit.full else None
    fit.dact[0].zero_(); fit.dact[1].zero_()          # the chain used the gradient maps as scratch (interiors are rewritten each step)
    if fit.conv_variant >= 4:
        # every fp32-accurate multiply-accumulate is 3 fp16 MFMA products (two error-compensated fp16 pieces per operand, fp32
        # accumulate): the pipe that bounds the kernel is the 16-bit matrix pipe at 1/3 of its dense peak
        peak = PEAK_BF16_MATRIX_TFLOPS / 3.0
        kname = ('conv3x3_pair_kernel (variant 5: TWO 64->64 layers per launch on 10x14 tiles, intermediate in LDS; two fp16 pieces per fp32 '
                 'operand, 3 products on v_mfma_f32_32x32x16_f16)' if pairs else
                 'conv3x3_wino_kernel (variant 10: ONE 64->64 layer per launch as Winograd F(2x2, 3x3): 32 tiles of 2x2 outputs per workgroup, fp32 '
                 'transforms, the 16 position GEMMs on two fp16 pieces per operand, 3 products on v_mfma_f32_32x32x16_f16)' if wino else
                 'conv3x3_split_kernel<NP=2> (variant 4: two fp16 pieces per fp32 operand, 3 products on v_mfma_f32_32x32x16_f16)')
        peak_note = ('algorithmic fp32 FLOP/s against f16 dense MFMA peak %.0f TF / 3 products per fp32-accurate MAC (variant 3, 6 bf16 '
                     'products: peak / 6; the fp32-MFMA kernel (--conv-variant 2): %.1f TF)' % (PEAK_BF16_MATRIX_TFLOPS, PEAK_FP32_MATRIX_TFLOPS))
        if wino:
            peak_note += ('.  ALGORITHMIC = the direct convolution\'s 2*H*W*64*576 FLOP per layer (SURVEY 8(d)); the Winograd form issues 16/36 of '
                          'those multiplies (issued f16-MFMA work: achieved * 3 * 4/9 TFLOP/s of the %.0f TF dense peak)' % PEAK_BF16_MATRIX_TFLOPS)
    elif fit.conv_variant == 3:
        # every fp32 multiply-accumulate is 6 bf16 MFMA products (exact 3-way operand split, fp32 accumulate):
        # the pipe that bounds the kernel is the bf16 matrix pipe at 1/6 of its dense peak
        peak, kname = PEAK_BF16_MATRIX_TFLOPS / 6.0, 'conv3x3_split_kernel (variant 3: fp32-exact 3xbf16 operand split on v_mfma_f32_32x32x16_bf16)'
        peak_note = ('algorithmic fp32 FLOP/s against bf16 dense MFMA peak %.0f TF / 6 products per fp32-exact MAC; the '
                     'fp32-MFMA kernel (--conv-variant 2) is priced against %.1f TF' % (PEAK_BF16_MATRIX_TFLOPS, PEAK_FP32_MATRIX_TFLOPS))
    else:
        peak, kname = PEAK_FP32_MATRIX_TFLOPS, f'conv3x3_mfma (variant {fit.conv_variant}, v_mfma_f32_32x32x2_f32)'
        peak_note = 'fp32-input MFMA peak'
    # algorithmic HBM bytes of the dominant launch: single layer = 8.4 MB in + 8.4 out + 0.3 weights; fused pair = in + out + the
    # intermediate map (forward: written as the saved activation; backward: read as the lrelu' operand) + the second operand map of
    # the backward epilogue (8.4 MB each) + 0.3 MB of weights: 25.5 MB forward, 33.9 MB backward, averaged over the 3 + 3 launches
    alg_bytes = (3 * 25.5e6 + 3 * 33.9e6) / 6.0 if pairs else 17.1e6
    out = {
        'metric': 'fitting-iterations/sec (T=120 frames)', 'value': world * args.steps / dt,
        'unit': 'fitting-iterations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'ramp_iterations': ramp_iters, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'opt_amass_temp.py temporal fit, one TotalCapture-shaped clip per GPU: B=119 frames '
                               '(the T=120 clip), SMPL-X-shaped synthetic model V=10475, VPoser decode, smoothness '
                               'encoder 245x134, marker+contact+prior losses, Adam',
                   'frames': B, 'vertices_per_frame': 10475 if not args.active_vertices_only else int(fit.n),
                   'synthetic_model': args.model + (' (skinning joints i.i.d. per vertex: every 512-vertex chunk touches all 55 joints -- the worst case)'
                                                     if args.model == 'iid' else ' (index locality of the licensed model: a 512-vertex chunk touches 4-17 joints)'),
                   'sequences': world, 'conv_variant': fit.conv_variant,
                   'arithmetic': arithmetic_label(fit.conv_variant),
                   'parallelism': f'seq-shard x{world} + 1 all_gather', 'hip_graph': use_graph},
        'final_total_loss': losses['total'],
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': achieved / peak, 'traffic': pmc_traffic('lemo::conv3x3_pair_kernel<0' if pairs else
                                                                         'lemo::conv3x3_wino_kernel<0' if wino else
                                                                         ('lemo::conv3x3_split_kernel<0, 64, 64' if fit.conv_variant >= 3
                                                                          else 'lemo::conv3x3_mfma_v2_kernel<0')),
                     'traffic_unit': 'bytes/launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction; algorithmic '
                                     'minimum %.1fe6)' % (alg_bytes / 1e6),
                     'peak_note': peak_note,
                     'kernel': kname + (' 64->64->64ch 245x134, 6 of the %d launches/iteration (12 of the 14 64->64 layers)' % {5: 25, 6: 25, 7: 23, 8: 22, 9: 21}.get(fit.conv_variant, 25) if pairs
                                        else ' 64->64ch 245x134, 14 of the 27 launches/iteration' if wino
                                        else ' 64->64ch 245x134, 14 of the 31 launches/iteration'),
                     'kernel_ms': kern_ms, 'flop_per_launch': kern_flops,
                     'kernel_ms_source': ('HIP events around a captured replay of the iteration\'s own chain of the six fused launches (3 forward '
                                          'pairs act[3] -> act[9], 3 backward-data pairs through the gradient maps, each reading what the previous '
                                          'one wrote, engine buffers), / 6' if pairs else
                                          'HIP events around a captured replay of the iteration\'s own chain of the fourteen 64->64 '
                                          'launches (7 fwd + 7 bwd-data, each reading what the previous one wrote, engine buffers), / 14') +
                                         ': the in-iteration duration incl. the kernel boundary, comparable with the rocprofv3 '
                                         'average of the same kernel in profiles/ (kernel stats of this round)',
                     'kernel_ms_back_to_back': b2b_ms,
                     'memory_view': {
                         'algorithmic_bytes_per_launch': alg_bytes, 'achieved_TBps': alg_bytes / (kern_ms * 1e-3) / 1e12,
                         'frac_of_hbm_peak': alg_bytes / (kern_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                         'copy_layer_floor_ms': 0.00507,
                         'frac_of_copy_layer_floor': 0.00507 / kern_ms,
                         'note': 'since round 3 the layer is bound by data movement + launch boundary, not by the matrix pipe (DESIGN 9.2): '
                                 'a kernel of the same launch geometry that only reads the 8.4 MB the previous launch wrote and writes 8.4 MB '
                                 'takes 5.07 us in a dependent chain (tools/ubench/boundary_ubench.hip, profiles/r03_boundary_ubench.txt); '
                                 'frac_of_copy_layer_floor = that floor / kernel_ms'},
                     'frac_back_to_back': kern_flops / (b2b_ms * 1e-3) / 1e12 / peak,
                     'traffic_source': 'committed PMC file ' + PMC_FILE + ' (counters cannot be read from inside the process)'},
    }
    if value_100 is not None:
        out['value_100_steps'] = value_100
    if stage_us is not None:
        # where one iteration's time goes (lemo_fit_census: every stage replayed 20 x back to back from its own graph, us per
        # iteration); 'forward_backward' is the whole chain measured the same way (no Adam update: + ~1 us in the tail launch)
        out['stage_us'] = dict({k: round(v, 2) for k, v in stage_us.items()}, sum_of_stages=round(sum(v for k, v in stage_us.items() if k != 'forward_backward'), 2),
                               ms_per_step_100=round(1e3 / value_100, 4) if value_100 else None)
    if vs is not None:
        vms_b2b, vbytes, vflops = vs
        vms = lbs_in_chain_ms if lbs_in_chain_ms and lbs_in_chain_ms > 0 else vms_b2b
        out['roofline']['hbm'] = {
            'kernel': 'lbs_verts_fwd_kernel (blend-shape GEMM 128 x 3V x 512 + skinning, all V = 10475 vertices x B frames; 1 launch/iteration)',
            'bound': 'hbm', 'achieved': vbytes / (vms * 1e-3) / 1e12, 'peak': PEAK_HBM_TBS, 'unit': 'TB/s',
            'frac': vbytes / (vms * 1e-3) / 1e12 / PEAK_HBM_TBS, 'kernel_ms': vms, 'bytes_per_launch': vbytes,
            'traffic': pmc_traffic('lemo::lbs_verts_fwd_kernel'), 'traffic_source': 'committed PMC file ' + PMC_FILE,
            'kernel_ms_source': 'replay of [lbs_verts_fwd + the 14-conv chain] minus replay of [the 14-conv chain]: the launch with '
                                'the caches in the state the encoder leaves them in (blend directions not MALL-hot)',
            'kernel_ms_back_to_back': vms_b2b,
            'blend_operands': 'two fp16 pieces each, pre-split (3 MFMA products per MAC)' if fit.dev.blend_f16 else 'three bf16 pieces each, split in the kernel (6 MFMA products per MAC)',
            'mfma_frac': vflops / (vms * 1e-3) / 1e12 / (PEAK_BF16_MATRIX_TFLOPS / (3.0 if fit.dev.blend_f16 else 6.0)), 'flop_per_launch': vflops}
    if per_rank is not None:
        out['per_rank_iterations_per_s'] = per_rank
    if rank == 0 and world == 1 and args.concurrent_clips > 1 and use_graph and not args.active_vertices_only:
        try:      # an extra, never the headline: a failure here must not cost the line
            out['concurrent_clips'] = concurrent_probe(fit, prob, B, device, args.concurrent_clips, max(args.steps, 100), args.conv_variant)
        except Exception as e:       # noqa: BLE001
            out['concurrent_clips'] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0 and world == 1 and not args.no_extras and use_graph and not args.active_vertices_only:
        # A/B rows the line carries itself: the same clip, same process, on the layer-by-layer split-f16 kernels (variant 4) and on
        # the fp32-input MFMA kernels (variant 2); then BASELINE's second metric, MPJPE of the full fit against the oracle's
        try:
            ab = {}
            if fit.conv_variant != 4:
                ab['value_layer_by_layer_f16x2'] = variant_probe(rank, B, device, stream, 4, fit)
            if fit.conv_variant != 8:
                ab['value_tail_without_layer_2'] = variant_probe(rank, B, device, stream, 8, fit)
            if fit.conv_variant != 7:
                ab['value_head_without_layer_2'] = variant_probe(rank, B, device, stream, 7, fit)
            if fit.conv_variant != 5:
                ab['value_pairs_without_fused_head_tail'] = variant_probe(rank, B, device, stream, 5, fit)
            if fit.conv_variant != 10:
                ab['value_winograd_layers'] = variant_probe(rank, B, device, stream, 10, fit)
            ab['value_fp32_mfma'] = variant_probe(rank, B, device, stream, 2, fit)
            ab['value_headline_again'] = max(timed_fit(fit, prob, stream, device) for _ in range(2))
            ab['note'] = ('fitting-iterations/s over 100 timed steps, same clip / process / box, interleaved: the encoder on conv variant 4 '
                          '(one split-f16 launch per layer), on variant 8 (tail = layers 1, 0 backwards; layer 2\'s backward a launch of its own), on variant 7 (head = marker image + layers 0, 1; layer 2 forward and backward launches of their own), on variant 5 (round 4\'s default: fused pairs, head and tail layer by layer), on variant 10 (round 6: every 64->64 layer one Winograd F(2x2, 3x3) launch instead of the fused pairs: 2.25 x fewer MFMAs, measured slower, '
                          'DESIGN 5), on variant 2 (v_mfma_f32_32x32x2_f32, fp32 operands) and the headline engine once more')
            out['variants'] = ab
        except Exception as e:       # noqa: BLE001
            out['variants'] = {'error': f'{type(e).__name__}: {e}'}
        try:
            out['mpjpe'] = mpjpe_probe(fit, prob, stream, device)
            out['mpjpe_mm'] = out['mpjpe']['value']
        except Exception as e:       # noqa: BLE001
            out['mpjpe'] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0 and world == 1 and not args.no_extras and use_graph and not args.active_vertices_only and args.model == 'iid':
        # the same two workloads on the synthetic model with the licensed model's index locality (the i.i.d.-joint model
        # of the headline is the worst case for the skinning gather and the chunked all-vertex LBS backward)
        try:
            from lemo_amd import synthetic as _syn
            _syn.DEFAULT_COHERENT = True
            try:
                fc, pc = build_problem(rank, B, device, full_vertices=True, conv_variant=fit.conv_variant)
                with torch.cuda.stream(stream):
                    fit.step(100, use_graph=True)
                torch.cuda.synchronize(device)
                v_co = max(timed_fit(fc, pc, stream, device) for _ in range(2))
                assert fc.nonfinite_step() == 0
                del fc, pc
                p_co = prox_probe(device)[1]['value']
            finally:
                _syn.DEFAULT_COHERENT = False
            out['coherent_model'] = {
                'value': v_co, 'prox_window_value': p_co, 'unit': 'fitting-iterations/s',
                'note': 'lemo_amd.synthetic.make_synthetic_smplx(coherent=True): consecutive vertex indices share their dominant joint in runs '
                        'of 20-400 vertices, a vertex is skinned to its part\'s joint and up to three tree neighbours (a 512-vertex chunk '
                        'touches 4-17 joints instead of all 55); same shapes and non-zero bounds as the headline\'s i.i.d. model.  The AMASS '
                        'iteration does not care (its skinning gather is LDS-issue bound either way); the PROX window (all-vertex LBS backward in '
                        '512-vertex chunks, SDF sampling) does'}
        except Exception as e:       # noqa: BLE001
            out['coherent_model'] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0 and world == 1 and not args.no_extras and use_graph:
        # the other workloads of BASELINE.json on this GPU (never the headline value; a failure must not cost the line)
        del fit
        for key, fn in (('prox_window', lambda: prox_probe(device)[1]), ('perframe', lambda: perframe_probe(device)),
                        ('ae_finetune', lambda: ae_probe(device))):
            try:
                out[key] = fn()
            except Exception as e:       # noqa: BLE001
                out[key] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(prob, B)
        out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
