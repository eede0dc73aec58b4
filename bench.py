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

Extra objects in the JSON line:
  roofline     the dominant kernel (largest device time among the single-symbol families, measured in the last warmup step with
               every family timed), re-timed alone with HIP events on the launch stream over the timed region: algorithmic FLOPs / time.
  whole_path   FLOPs per image as the reference computes them and as the kernels execute them (the decoder's Upsample2x convs run
               in a folded 4-tap form), both as a fraction of the MFMA peak of the mode; mfma_time_weighted: FLOPs / time summed over
               every GEMM / conv / attention launch of the profiled warmup step.
  cpu_baseline the CPU oracle (oracle/, a scalar C port of the reference algorithm; kind "port") timed on this box's host
               cores on a bounded sample: four images through all 10 scales + decode (rank 0, N=1 only).
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
MFMA_FAMILIES = ('gemm', 'gemm_small', 'conv3x3', 'conv_small', 'attn')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU (weak scaling)')
    ap.add_argument('--depth', type=int, default=16)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rng-mode', default='exact', choices=['exact', 'per_rank'])
    ap.add_argument('--host-init', action='store_true', help='generate the detinit weights with numpy on the host (same bits; keeps the ~8000 tiny init kernels out of a rocprofv3 counter pass)')
    ap.add_argument('--kernel-breakdown', action='store_true',
                    help='time every kernel family with HIP events in the timed region too (adds ~2 %% to a step); default: only the dominant kernel')
    args = ap.parse_args()

    # `python bench.py --gpus N` outside a torchrun environment: start the N ranks ourselves.  Nothing above or below this point in
    # THIS process touches a GPU — the ranks are children (python -m torch.distributed.run), rank 0 prints the JSON line.
    from var_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    from var_amd import dist, hip
    from var_amd import detinit
    from var_amd.multi import sample_sharded

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):      # under torchrun, a 1-rank launch too (exercises RCCL)
        dist.initialize(backend='nccl')
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
    var.set_hip_precision(args.dtype)

    B_local, B_total = args.batch, args.batch * world
    labels = ((torch.arange(B_total) * 7) % 1000).to(dev)

    def step(i):
        return sample_sharded(var, B_total, labels, g_seed=i, cfg=1.5, top_k=900, top_p=0.96, rng_mode=args.rng_mode, gather=True)

    # warmup; the last warmup step runs with every family timed: it names the dominant kernel and gives the per-family table
    prepass = None
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if last: hip.timing_reset(); hip.timing_enable(True, None)
        step(i)
        if last:
            torch.cuda.synchronize(); hip.timing_enable(False); prepass = hip.timing_read()
    if prepass is not None:
        dominant = max(('gemm', 'conv3x3', 'attn'), key=lambda k: prepass[k]['ms'])       # the families that map to one kernel symbol each
    else:
        dominant = 'conv3x3' if (args.depth <= 16 and args.dtype == 'f32') else 'gemm'    # (--warmup 0: r01/r02 profiles)
    hip.timing_reset(); hip.timing_enable(True, None if args.kernel_breakdown else [dominant])
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        img = step(1000 + i)
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    hip.timing_enable(False)
    tt = hip.timing_read()
    assert img.shape == (B_total, 3, 256, 256) and bool(torch.isfinite(img).all())
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        f16 = args.dtype == 'f16'
        ips = B_total * args.steps / dt
        eng = var.engine()
        flops_img = eng.flops_per_image()
        dec_ref = eng.dec.flops_per_image_reference(pns[-1])        # as the reference computes the decoder (9-tap upsample convs)
        dec_exec = eng.dec.flops_per_image_executed(pns[-1])        # as the kernels execute it (folded 4-tap upsample convs)
        fam_peak = {k: (PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS) for k in MFMA_FAMILIES}
        f = tt[dominant]
        achieved = f['flops'] / (f['ms'] * 1e-3) / 1e12 if f['ms'] > 0 else 0.0
        kname = {('gemm', False): 'k_dma_gemm<4,4,false,2,false>', ('conv3x3', False): 'k_dma_gemm<4,5,true,2,false>', ('attn', False): 'k_attn_cached<4>',
                 ('gemm', True): 'k_gemm16<8,4,2,4>', ('conv3x3', True): 'k_conv16h<5,32>', ('attn', True): 'k_attn16<4>'}[(dominant, f16)]
        peak = fam_peak[dominant]
        traffic, tsrc = None, None                                   # HBM-side bytes per launch from a separate rocprofv3 --pmc pass of the same config
        for prof, pdt in ((f'r02_{args.dtype}_pmc_traffic.json', args.dtype), ('r01_pmc_traffic.json', 'f32')):
            try:
                pj = json.load(open(os.path.join(ROOT, 'profiles', prof)))
                pm = pj['kernels'].get(kname)
                if pm and args.batch == 64 and args.depth == 16 and pdt == args.dtype:
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
            ms = sum(table[k]['ms'] for k in MFMA_FAMILIES); fl = sum(table[k]['flops'] for k in MFMA_FAMILIES)
            ideal_ms = sum(table[k]['flops'] / (fam_peak[k] * 1e9) for k in MFMA_FAMILIES)
            whole['mfma_time_weighted'] = {'tflops': round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else None,
                                           'frac_of_peak': round(ideal_ms / ms, 4) if ms > 0 else None,
                                           'device_ms_per_step': round(ms / (args.steps if args.kernel_breakdown else 1), 3),
                                           'source': 'timed region' if args.kernel_breakdown else 'last warmup step (every family timed)'}
        out = {
            'metric': '256x256 images/sec (CFG=1.5) VAR-d%d' % args.depth, 'value': round(ips, 3), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'VAR-d{args.depth} 256x256 full 10-scale pyramid, CFG=1.5, top_k=900, top_p=0.96, batch={B_local}/GPU, random-init (detinit seed 0)',
                       'global_batch': B_total, 'parallelism': f'dp{world} (batch shard, RCCL all-gather of decoded images)', 'rng_mode': args.rng_mode,
                       'precision': 'fp32 parity mode' if not f16 else 'fp16 GEMM / conv operands, activations and KV cache, fp32 accumulate and statistics'},
            'roofline': {'bound': 'mfma', 'kernel': kname, 'dominant_by': 'measured (last warmup step)' if prepass is not None else 'profile',
                         'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                         'traffic': traffic, 'traffic_source': tsrc,
                         'launches': f['launches'], 'avg_launch_ms': round(f['ms'] / max(f['launches'], 1), 5),
                         'algorithmic_gflop_per_launch': round(f['flops'] / max(f['launches'], 1) / 1e9, 3),
                         'algorithmic_mbytes_per_launch': round(f['bytes'] / max(f['launches'], 1) / 1e6, 3)},
            'kernel_time_ms_per_step': {k: round(v['ms'] / args.steps, 3) for k, v in tt.items() if v['launches'] > 0},
            'kernel_tflops': {k: round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) for k, v in tt.items() if v['ms'] > 0 and v['flops'] > 0},
            'kernel_algorithmic_gbps': {k: round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) for k, v in tt.items() if v['ms'] > 0 and v['bytes'] > 0},
            'peak_hbm_allocated_gib': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            'whole_path': whole,
        }
        if prepass is not None:
            out['warmup_step_kernel_ms'] = {k: round(v['ms'], 3) for k, v in prepass.items() if v['launches'] > 0}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.depth, pns)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.finalize()


def cpu_baseline(depth, pns):
    """the CPU oracle on a bounded sample — FOUR images (about 13 s), all scales + decode — OpenMP over this box's host cores"""
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
    nimg = 4
    noise = [torch.empty(nimg * pn * pn, 4096).exponential_(1, generator=g).numpy() for pn in pns]
    t0 = time.perf_counter()
    r = orc.run([7, 14, 21, 28][:nimg], noise, 1.5, 900, 0.96)
    dt = time.perf_counter() - t0
    assert np.isfinite(r['img']).all()
    return {'value': round(nimg / dt, 4), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{nimg} images (labels 7, 14, 21, 28), all 10 scales + VQVAE decode, oracle/var_oracle.c via OpenMP: {dt:.1f} s'}


if __name__ == '__main__':
    main()
