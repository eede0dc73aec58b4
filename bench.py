#!/usr/bin/env python3
"""bench.py — 256x256 images/sec of VAR-d16 next-scale sampling (CFG 1.5, top-k 900, top-p 0.96) on N MI355X.

One "step" = one VAR.autoregressive_infer_cfg-equivalent call: the whole 10-scale sampling loop + VQVAE decode for
B=64 images per GPU (BASELINE.json configs[1]), random-init weights (var_amd.detinit, seed 0), labels (i*7) mod 1000.
N>1: one rank per GPU over RCCL — under torchrun, or started by this script itself when `--gpus N` is given outside a torchrun
environment (var_amd/launch.py; the parent never touches a GPU).  The batch is sharded image-wise, no collective inside the loop,
one all-gather of the decoded images inside the timed region (var_amd/multi.py).  `value` = images of all ranks / max-over-ranks time.

--dtype f32 (default, the driver's line): the parity mode — token ids bit-identical to the CPU oracle, priced against the 157.3 TF
fp32 MFMA peak.  --dtype f16: the 16-bit throughput mode of the transformer (fp16 GEMM operands / KV cache, fp32 accumulation; what the
reference's harness requests with torch.autocast(fp16), demo_sample.py:66-68), priced against the 2.5 PF dense fp16 MFMA peak.
--dtype bf16: the same mode with bfloat16 storage (the reference's other 16-bit option, utils/arg_util.py `fp16: int  # 1: using fp16, 2: bf16`).

Extra objects in the JSON line:
  modes.f16    (default run, after the f32 timed region, same process / weights / workload) the 16-bit throughput mode: 2 warmup + >= 10 timed
               steps between barriers, `value`, `ms_per_step`, its own measured dominant kernel with `roofline` and `whole_path`.  The
               headline `value` / `dtype` stay f32.
  modes.bf16   the same for the bfloat16 flavour of that mode.
               Both carry `two_calls_in_flight` at N = 1 (at N > 1 only with --in-flight): the same K calls issued alternately on two HIP streams
               (informational: a serving loop with two requests in flight; every `value` / `ms_per_step` above it is one call at a time on one stream).
               At N > 1 a mode that raises is reported as modes.<dtype>.error and the headline line is still printed.
  ranks        (N > 1) each rank's own wall time per step, min and max over ranks, and the HIP-event time of the RCCL all-gather.
  roofline     the dominant kernel (largest device time among the single-symbol families, measured in the last warmup step with
               every family timed), re-timed alone with HIP events on the launch stream over the timed region: algorithmic FLOPs / time.
  whole_path   FLOPs per image as the reference computes them and as the kernels execute them (the decoder's Upsample2x convs run
               in a folded 4-tap form), both as a fraction of the MFMA peak of the mode; mfma_time_weighted: FLOPs / time summed over
               every GEMM / conv / attention launch of the profiled warmup step.
  cpu_baseline the CPU oracle (oracle/, a scalar C port of the reference algorithm; kind "port") timed on this box's host
               cores on a bounded sample: eight images through all 10 scales + decode (rank 0, N=1 only).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4, 64 FLOP/clk/SIMD
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense BF16/FP16 MFMA
PEAK_HBM_GBS = 8000.0
# MFMA kernel families of the library's timing table (include/var_hip.h) and the peak each is priced against: a family holds kernels of
# ONE arithmetic type.  In the 16-bit mode the fp32 families still run (AdaLN / head_nm projections, the decoder's four attention blocks).
FAMILY_PEAK = {'gemm': PEAK_F32_MFMA_TFLOPS, 'gemm_small': PEAK_F32_MFMA_TFLOPS, 'conv3x3': PEAK_F32_MFMA_TFLOPS, 'conv_small': PEAK_F32_MFMA_TFLOPS,
               'attn': PEAK_F32_MFMA_TFLOPS, 'gemm16': PEAK_F16_MFMA_TFLOPS, 'gemm16_small': PEAK_F16_MFMA_TFLOPS, 'conv16h': PEAK_F16_MFMA_TFLOPS,
               'conv16_small': PEAK_F16_MFMA_TFLOPS, 'attn16': PEAK_F16_MFMA_TFLOPS}
# families that can dominate a step -> the kernel symbol behind them (gemm / conv3x3 / gemm16 (the persistent 256x256 kernel): exactly one symbol; conv16h: the halo-patch
# kernel at 160 output channels, GroupNorm-fused (<5,32,true>: all but one launch of a decode) or plain (<5,32,false>); the attention families: one template, 1-4 waves per
# workgroup by l)
DOMINANT = {'f32': {'gemm': 'k_dma_gemm<4,4,false,2,false>', 'conv3x3': 'k_dma_gemm<4,5,true,2,false>', 'attn': 'k_attn_cached<NW>'},
            'f16': {'gemm16': 'k_gemm16p', 'conv16h': 'k_conv16h<5,32,true>', 'attn16': 'k_attn16<NW>'}}
DOMINANT['bf16'] = DOMINANT['f16']          # the same kernels compiled with the bf16 MFMA opcodes (namespace vh_bf16 in the symbol)


_STREAMS = []


def _two_streams(torch):
    """the same two side streams for every mode (HIP maps streams onto a few hardware queues: a fresh pair per mode can land on one queue)"""
    if not _STREAMS:
        _STREAMS.extend([torch.cuda.Stream(), torch.cuda.Stream()])
    return _STREAMS


def run_mode(var, dtype, steps, warmup, args, world, step_fn, dist, hip, torch, in_flight2=False):
    """warm up and time `steps` calls in one precision mode; returns the raw measurements of this rank (dt already MAX-reduced over ranks)"""
    var.set_hip_precision(dtype)
    prepass = None
    for i in range(warmup):
        last = i == warmup - 1
        if last: hip.timing_reset(); hip.timing_enable(True, None)
        step_fn(i, None)
        if last:
            torch.cuda.synchronize(); hip.timing_enable(False); prepass = hip.timing_read()
    fams = DOMINANT[dtype]
    if prepass is not None:
        dominant = max(fams, key=lambda k: prepass[k]['ms'])
    else:
        dominant = {'f32': 'conv3x3' if args.depth <= 16 else 'gemm', 'f16': 'gemm16', 'bf16': 'gemm16'}[dtype]          # (--warmup 0)
    hip.timing_reset(); hip.timing_enable(True, None if args.kernel_breakdown else [dominant])
    gather_ev = []
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        img = step_fn(1000 + i, gather_ev)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                       # this rank's own time for the K steps (before waiting for the others)
    dist.barrier()
    dt = time.perf_counter() - t0
    hip.timing_enable(False)
    tt = hip.timing_read()
    B_total = args.batch * world
    assert img.shape[0] == B_total and img.shape[1] == 3 and bool(torch.isfinite(img).all())
    gather_ms = sum(a.elapsed_time(b) for a, b in gather_ev) / max(steps, 1)
    ranks = None
    if world > 1:
        from var_amd.multi import rank_stats
        rs = rank_stats(dt, dt_own, gather_ms, steps, img.shape, img.device)
        dt, ranks = rs['dt_max'], rs['ranks']
    res = dict(dt=dt, tt=tt, prepass=prepass, dominant=dominant, ranks=ranks, steps=steps, warmup=warmup, dtype=dtype)
    if in_flight2:
        # the same K calls again, issued alternately on two HIP streams (a call's buffers are per stream): two calls in flight, so that the
        # latency-bound small scales and ragged last rounds of one call run beside the other's full-chip kernels.  Same barriers, same clock.
        streams = _two_streams(torch)
        for i in range(2):
            with torch.cuda.stream(streams[i]): step_fn(500 + i, None)
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % 2]): img = step_fn(1000 + i, None)
        torch.cuda.synchronize()
        dist.barrier()
        dt2 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt2], dtype=torch.float64, device=dist.collective_device(img.device))
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt2 = float(t.item())
        res['dt_in_flight2'] = dt2
    return res


def describe_mode(m, args, world, var, pns):
    """rank 0: the numbers of one precision mode as the JSON objects of the bench line"""
    dtype, dt, tt, prepass, dominant, steps = m['dtype'], m['dt'], m['tt'], m['prepass'], m['dominant'], m['steps']
    f16 = dtype != 'f32'
    B_total = args.batch * world
    ips = B_total * steps / dt
    eng = var.engine()
    flops_img = eng.flops_per_image()
    dec_ref = eng.dec.flops_per_image_reference(pns[-1])        # as the reference computes the decoder (9-tap upsample convs)
    dec_exec = eng.dec.flops_per_image_executed(pns[-1])        # as the kernels execute it (folded 4-tap upsample convs)
    f = tt[dominant]
    achieved = f['flops'] / (f['ms'] * 1e-3) / 1e12 if f['ms'] > 0 else 0.0
    kname = DOMINANT[dtype][dominant]
    peak = FAMILY_PEAK[dominant]
    traffic, tsrc = None, None                                   # HBM-side bytes per launch from a separate rocprofv3 --pmc pass of the same config
    for prof in (f'r04_{dtype}_pmc_traffic.json',) + (() if dtype == 'bf16' else (f'r03_{dtype}_pmc_traffic.json', f'r02_{dtype}_pmc_traffic.json')):
        try:
            pj = json.load(open(os.path.join(ROOT, 'profiles', prof)))
            pm = pj['kernels'].get(kname)
            if pm and args.batch == 64 and args.depth == 16:
                traffic, tsrc = pm['traffic_bytes_per_launch'], f'profiles/{prof} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)'
                break
        except (OSError, KeyError, ValueError):
            pass
    whole_peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
    whole = {'gflop_per_image_reference': round((flops_img + dec_ref) / 1e9, 1), 'gflop_per_image_executed': round((flops_img + dec_exec) / 1e9, 1),
             'tflops_reference': round(ips * (flops_img + dec_ref) / 1e12 / world, 2), 'tflops_executed': round(ips * (flops_img + dec_exec) / 1e12 / world, 2),
             'frac_of_mfma_peak_reference_flops': round(ips * (flops_img + dec_ref) / 1e12 / world / whole_peak, 4),
             'frac_of_mfma_peak_executed_flops': round(ips * (flops_img + dec_exec) / 1e12 / world / whole_peak, 4),
             'peak_tflops': whole_peak,
             'note': 'ada_lin counted once per call (hoisted; the reference recomputes it per scale)'}
    table = tt if args.kernel_breakdown else prepass
    if table is not None:
        fams = [k for k in FAMILY_PEAK if table[k]['launches'] > 0]
        ms = sum(table[k]['ms'] for k in fams); fl = sum(table[k]['flops'] for k in fams)
        ideal_ms = sum(table[k]['flops'] / (FAMILY_PEAK[k] * 1e9) for k in fams)          # every family against the peak of ITS arithmetic type
        whole['mfma_time_weighted'] = {'tflops': round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else None,
                                       'frac_of_peak': round(ideal_ms / ms, 4) if ms > 0 else None,
                                       'device_ms_per_step': round(ms / (steps if args.kernel_breakdown else 1), 3),
                                       'families': fams,
                                       'source': 'timed region' if args.kernel_breakdown else 'last warmup step (every family timed)'}
    roof = {'bound': 'mfma', 'kernel': kname, 'family': dominant, 'dominant_by': 'measured (last warmup step)' if prepass is not None else 'profile',
            'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
            'traffic': traffic, 'traffic_source': tsrc,
            'launches': f['launches'], 'avg_launch_ms': round(f['ms'] / max(f['launches'], 1), 5),
            'algorithmic_gflop_per_launch': round(f['flops'] / max(f['launches'], 1) / 1e9, 3),
            'algorithmic_mbytes_per_launch': round(f['bytes'] / max(f['launches'], 1) / 1e6, 3)}
    extra = {'kernel_time_ms_per_step': {k: round(v['ms'] / steps, 3) for k, v in tt.items() if v['launches'] > 0},
             'kernel_tflops': {k: round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) for k, v in tt.items() if v['ms'] > 0 and v['flops'] > 0},
             'kernel_algorithmic_gbps': {k: round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) for k, v in tt.items() if v['ms'] > 0 and v['bytes'] > 0}}
    if prepass is not None:
        extra['warmup_step_kernel_ms'] = {k: round(v['ms'], 3) for k, v in prepass.items() if v['launches'] > 0}
    return ips, roof, whole, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU (weak scaling)')
    ap.add_argument('--depth', type=int, default=16)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16'], help='the mode of the headline value (default f32: the parity contract)')
    ap.add_argument('--no-modes', action='store_true', help='skip the extra 16-bit-mode measurements that follow the f32 timed region (modes.f16, modes.bf16)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--in-flight', action='store_true',
                    help='N > 1 only: also run the informational two-calls-in-flight leg of modes.* (at N = 1 it always runs); it issues the all-gathers '
                         'alternately from two side streams, a pattern no multi-GPU run has exercised, so a scaling run leaves it out by default')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help=argparse.SUPPRESS)   # gloo: rehearsal of the N > 1 code path with ranks sharing one card (tests)
    ap.add_argument('--rng-mode', default='exact', choices=['exact', 'per_rank'])
    ap.add_argument('--host-init', action='store_true', help='generate the detinit weights with numpy on the host (same bits; keeps the ~8000 tiny init kernels out of a rocprofv3 counter pass)')
    ap.add_argument('--kernel-breakdown', action='store_true',
                    help='time every kernel family with HIP events in the timed region too (adds ~2 %% to a step); default: only the dominant kernel')
    args = ap.parse_args()

    # `python bench.py --gpus N` outside a torchrun environment: start the N ranks ourselves.  Nothing above or below this point in
    # THIS process touches a GPU — the ranks are children (python -m torch.distributed.run), rank 0 prints the JSON line.
    from var_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, need_gpus=args.backend == 'nccl'))

    import torch
    from var_amd import dist, hip
    from var_amd import detinit
    from var_amd.multi import sample_sharded

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):      # under torchrun, a 1-rank launch too (exercises RCCL)
        dist.initialize(backend=args.backend)
    else:
        torch.cuda.set_device(0)
    rank = dist.get_rank()
    dev = torch.device('cuda', torch.cuda.current_device())
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus}, or without torchrun'

    from models import build_vae_var
    pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
    with contextlib.redirect_stdout(io.StringIO()):
        vae, var = build_vae_var(device=dev, patch_nums=pns, depth=args.depth, ch=160)
    fill_module_ = detinit.fill_module_ if args.host_init else detinit.fill_module_device_      # identical bits (tests/test_host_cpu.py); the device form takes seconds
    fill_module_(var, args.depth, 0, 'var.'); fill_module_(vae, args.depth, 0, 'vae.')
    var.eval(); vae.eval()
    var.rng = torch.Generator(device=dev)

    B_local, B_total = args.batch, args.batch * world
    labels = ((torch.arange(B_total) * 7) % 1000).to(dev)

    def step(i, gather_ev):
        return sample_sharded(var, B_total, labels, g_seed=i, cfg=1.5, top_k=900, top_p=0.96, rng_mode=args.rng_mode, gather=True, gather_events=gather_ev)

    # the headline mode: W warmup steps (the last one with every family timed: it names the dominant kernel), then exactly K timed steps
    head = run_mode(var, args.dtype, args.steps, args.warmup, args, world, step, dist, hip, torch)
    # then, in the same process, the 16-bit throughput mode (what the reference's harness runs under torch.autocast(fp16)): its own warmup,
    # >= 10 timed steps, its own dominant kernel and roofline.  The headline value / dtype stay those of the parity mode.
    # N > 1: one stream, one all-gather per step as in the headline region (the two-stream leg only with --in-flight), and a failure here is
    # reported inside `modes` instead of costing the headline line (every rank runs the same code, so a raise is a raise on all of them).
    others, mode_errors = {}, {}
    if args.dtype == 'f32' and not args.no_modes:
        for dt16 in ('f16', 'bf16'):
            try:
                others[dt16] = run_mode(var, dt16, max(10, args.steps), 2, args, world, step, dist, hip, torch, in_flight2=(world == 1 or args.in_flight))
            except Exception as e:          # noqa: BLE001
                if world == 1: raise
                mode_errors[dt16] = f'{type(e).__name__}: {e}'

    if rank == 0:
        precision = {'f32': 'fp32 parity mode', 'f16': 'fp16 GEMM / conv operands, activations and KV cache, fp32 accumulate and statistics',
                     'bf16': 'bfloat16 GEMM / conv operands, activations and KV cache, fp32 accumulate and statistics'}
        var.set_hip_precision(args.dtype)
        ips, roof, whole, extra = describe_mode(head, args, world, var, pns)
        out = {
            'metric': '256x256 images/sec (CFG=1.5) VAR-d%d' % args.depth, 'value': round(ips, 3), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(head['dt'] / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'VAR-d{args.depth} 256x256 full 10-scale pyramid, CFG=1.5, top_k=900, top_p=0.96, batch={B_local}/GPU, random-init (detinit seed 0)',
                       'global_batch': B_total, 'parallelism': f'dp{world} (batch shard, RCCL all-gather of decoded images)', 'rng_mode': args.rng_mode,
                       'precision': precision[args.dtype]},
            'roofline': roof, **extra,
            'peak_hbm_allocated_gib': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            'whole_path': whole,
        }
        if head['ranks'] is not None:
            out['ranks'] = head['ranks']
        for dt16, other in others.items():
            var.set_hip_precision(dt16)
            ips2, roof2, whole2, extra2 = describe_mode(other, args, world, var, pns)
            out.setdefault('modes', {})[dt16] = {'value': round(ips2, 3), 'unit': 'images/sec', 'ms_per_step': round(other['dt'] / other['steps'] * 1e3, 3),
                                                 'steps': other['steps'], 'warmup': other['warmup'], 'dtype': dt16, 'precision': precision[dt16],
                                                 'same_workload_as_headline': True, 'speedup_vs_headline': round(ips2 / ips, 3),
                                                 'roofline': roof2, 'whole_path': whole2, **extra2}
            if other['ranks'] is not None:
                out['modes'][dt16]['ranks'] = other['ranks']
            if 'dt_in_flight2' in other:
                v2 = B_total * other['steps'] / other['dt_in_flight2']
                out['modes'][dt16]['two_calls_in_flight'] = {
                    'value': round(v2, 3), 'unit': 'images/sec', 'ms_per_step': round(other['dt_in_flight2'] / other['steps'] * 1e3, 3), 'steps': other['steps'],
                    'speedup_vs_one_call_at_a_time': round(v2 / ips2, 3),
                    'frac_of_mfma_peak_reference_flops': round(whole2['frac_of_mfma_peak_reference_flops'] * v2 / ips2, 4),
                    'how': 'the same calls issued alternately on two HIP streams of one process and model (per-stream workspaces); `value` above is one call at a time'}
        for dt16, err in mode_errors.items():
            out.setdefault('modes', {})[dt16] = {'error': err}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.depth, pns)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.finalize()


def cpu_baseline(depth, pns):
    """the CPU oracle on a bounded sample — EIGHT images (about 10 s on 16 cores), all scales + decode — OpenMP over this box's host cores"""
    import numpy as np
    import torch
    from oracle.var_oracle import OracleVAR
    from var_amd import shapes
    from var_amd.detinit import make_state_dict
    var_sd = make_state_dict(shapes.var_shapes(depth, pns), depth=depth, seed=0, prefix='var.')
    var_sd['lvl_1L'] = np.concatenate([np.full((p * p,), i, dtype=np.int64) for i, p in enumerate(pns)]).reshape(1, -1)
    vae_sd = make_state_dict(shapes.vae_shapes(ch=160, patch_nums=pns, include_encoder=False), depth=depth, seed=0, prefix='vae.')
    orc = OracleVAR(var_sd, vae_sd, pns, depth)
    import ctypes
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    so = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libvar_oracle.so'))
    threads = int(so.varref_set_threads(int(os.environ.get('OMP_NUM_THREADS', min(avail, 16)))))   # a 1-GPU box's CPU share is 16 cores
    g = torch.Generator().manual_seed(0)
    nimg = 8
    noise = [torch.empty(nimg * pn * pn, 4096).exponential_(1, generator=g).numpy() for pn in pns]
    t0 = time.perf_counter()
    labels = [7 * (i + 1) for i in range(nimg)]
    r = orc.run(labels, noise, 1.5, 900, 0.96)
    dt = time.perf_counter() - t0
    assert np.isfinite(r['img']).all()
    return {'value': round(nimg / dt, 4), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{nimg} images (labels 7, 14, ..., {7 * nimg}), all 10 scales + VQVAE decode, oracle/var_oracle.c via OpenMP: {dt:.1f} s'}


if __name__ == '__main__':
    main()
