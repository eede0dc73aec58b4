#!/usr/bin/env python3
"""bench.py — 256x256 images/sec of VAR-d16 next-scale sampling (CFG 1.5, top-k 900, top-p 0.96) on N MI355X.

One "step" = one VAR.autoregressive_infer_cfg-equivalent call: the whole 10-scale sampling loop + VQVAE decode for
B=64 images per GPU (BASELINE.json configs[1]), random-init weights (var_amd.detinit, seed 0), labels (i*7) mod 1000.
N>1 (torchrun, one rank per GPU over RCCL): the batch is sharded image-wise, no collective inside the loop, one all-gather of
the decoded images inside the timed region (var_amd/multi.py).  `value` = images of all ranks / max-over-ranks time.

Extra objects in the JSON line:
  roofline     the dominant kernel family (by device time) of the timed region, measured with HIP events on the launch
               stream by the library's timing table (include/var_hip.h): algorithmic FLOPs / time vs the fp32 MFMA peak.
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

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU (weak scaling)')
    ap.add_argument('--depth', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rng-mode', default='exact', choices=['exact', 'per_rank'])
    ap.add_argument('--kernel-breakdown', action='store_true',
                    help='time every kernel family with HIP events (adds ~2 %% to a step); default: only the dominant kernel of the roofline object')
    args = ap.parse_args()

    # `python bench.py --gpus N` outside a torchrun environment: start the N ranks ourselves.  Nothing above or below this point in
    # THIS process touches a GPU — the ranks are children (python -m torch.distributed.run), rank 0 prints the JSON line.
    from var_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    from var_amd import dist, hip
    from var_amd.detinit import fill_module_device_ as fill_module_      # the detinit values, computed on the GPU (bit-identical to the numpy generator)
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
    fill_module_(var, args.depth, 0, 'var.'); fill_module_(vae, args.depth, 0, 'vae.')
    var.eval(); vae.eval()
    var.rng = torch.Generator(device=dev)

    B_local, B_total = args.batch, args.batch * world
    labels = ((torch.arange(B_total) * 7) % 1000).to(dev)

    def step(i):
        return sample_sharded(var, B_total, labels, g_seed=i, cfg=1.5, top_k=900, top_p=0.96, rng_mode=args.rng_mode, gather=True)

    for i in range(args.warmup):
        step(i)
    # HIP events around the launches, on the launch stream: every family with --kernel-breakdown, else only the dominant kernel.
    # profiles/r01_bench_kernel_stats.csv: at d16 the decoder's 128x160 implicit-GEMM conv instantiation leads (33 % of the device
    # time; the transformer GEMMs are spread over four tile instantiations of the same kernel, the largest at 21 %); at d30 (measured: 28 %)
    # the 128x128 transformer GEMM instantiation leads; depths in between are assumed to follow d30.
    dominant = 'conv3x3' if args.depth <= 16 else 'gemm'
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
        ips = B_total * args.steps / dt
        flops_img = var.engine().flops_per_image()
        dec_flops_img = var.engine().dec.flops_per_image_reference(pns[-1])          # as the reference computes the decoder (9-tap upsample convs)
        # dominant kernel by device time: with the full breakdown the measured leader among the single-symbol families is picked
        fam = max(('gemm', 'conv3x3', 'attn'), key=lambda k: tt[k]['ms']) if args.kernel_breakdown else dominant
        f = tt[fam]
        achieved = f['flops'] / (f['ms'] * 1e-3) / 1e12 if f['ms'] > 0 else 0.0
        kname = {'gemm': 'k_dma_gemm<4,4,false,2,false>', 'conv3x3': 'k_dma_gemm<4,5,true,2,false>', 'attn': 'k_attn_cached'}[fam]
        traffic = None                                                       # HBM-side bytes per launch from a separate rocprofv3 --pmc pass
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))['kernels'].get(kname)
            if pm and args.batch == 64 and args.depth == 16: traffic = pm['traffic_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        out = {
            'metric': '256x256 images/sec (CFG=1.5) VAR-d%d' % args.depth, 'value': round(ips, 3), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'VAR-d{args.depth} 256x256 full 10-scale pyramid, CFG=1.5, top_k=900, top_p=0.96, batch={B_local}/GPU, random-init (detinit seed 0)',
                       'global_batch': B_total, 'parallelism': f'dp{world} (batch shard, RCCL all-gather of decoded images)', 'rng_mode': args.rng_mode},
            'roofline': {'bound': 'mfma', 'kernel': kname,
                         'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         'traffic': traffic, 'traffic_source': 'profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)' if traffic else None,
                         'launches': f['launches'], 'avg_launch_ms': round(f['ms'] / max(f['launches'], 1), 5),
                         'algorithmic_gflop_per_launch': round(f['flops'] / max(f['launches'], 1) / 1e9, 3),
                         'algorithmic_mbytes_per_launch': round(f['bytes'] / max(f['launches'], 1) / 1e6, 3)},
            'kernel_time_ms_per_step': {k: round(v['ms'] / args.steps, 3) for k, v in tt.items() if v['launches'] > 0},
            'kernel_tflops': {k: round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) for k, v in tt.items() if v['ms'] > 0 and v['flops'] > 0},
            'kernel_algorithmic_gbps': {k: round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) for k, v in tt.items() if v['ms'] > 0 and v['bytes'] > 0},
            'peak_hbm_allocated_gib': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            'whole_path': {'gflop_per_image': round((flops_img + dec_flops_img) / 1e9, 1),
                           'tflops': round(ips * (flops_img + dec_flops_img) / 1e12 / world, 2),
                           'frac_of_f32_mfma_peak': round(ips * (flops_img + dec_flops_img) / 1e12 / world / PEAK_F32_MFMA_TFLOPS, 4)},
        }
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
