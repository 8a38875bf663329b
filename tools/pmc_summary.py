#!/usr/bin/env python3
"""Collate rocprofv3 --pmc passes (one output dir per counter group, csv format) into a text
summary and profiles/pmc_traffic.json (HBM/fabric bytes per launch, keyed by C-ABI entry point).
usage: pmc_summary.py <glob of p_counter_collection.csv> <out.txt> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

LAT = [int(i) for i in os.environ.get('L2Q_KPROF_LATTICE', '8 8 8 8').split()]
NB = int(os.environ.get('L2Q_KPROF_NB', 256))

ENTRY = {   # kernel-name fragment -> C-ABI entry point whose roofline it feeds
    'su3_plaq_slice_kernel': 'l2q_su3_plaq_reduce', 'su3_plaq_kernel': None, 'su3_plaq_sweep_kernel': None,
    'su3_force_slice_kernel<false': 'l2q_su3_force', 'su3_force_tile_kernel<false': None,
    'su3_force_kernel<false': None, 'su3_force_rows_kernel<0': None, 'su3_force_nu_kernel<0': None, 'su3_force_nu_kernel<1': None,
    'su3_force_plaq_kernel': 'l2q_su3_force',
    'su3_force_link_kernel<0': 'l2q_su3_force', 'su3_force_link_kernel<1': 'l2q_su3_force_kick',
    'su3_force_brick_kernel<0': 'l2q_su3_force', 'su3_force_brick_kernel<1': 'l2q_su3_force_kick',
    'su3_force_rows_kernel<1': None,
    'su3_force_slice_kernel<true': 'l2q_su3_force_kick',
    'su3_expm_mul_kernel<true, true>': 'l2q_su3_expm_mul2_vec8',
    'fused_heads_dma_kernel<true, true, true, false>': 'l2q_vnet_heads_vupdate_pair_f64',
    'fused_heads_dma_kernel<true, true, false, false>': 'l2q_vnet_heads_vupdate_f64',
    'gemm_dma_f64_kernel<false, false, false>': 'l2q_gemm_f64',
    'heads_sliced_kernel<true, true, true, false>': 'l2q_vnet_heads_vupdate_sliced_f64',
    'heads_sliced_kernel<true, true, true, false, false>': 'l2q_vnet_heads_vupdate_sliced_f64',
    # the reverse sweep (tools/kprof_train.py)
    'heads_sliced_kernel<true, true, false, false, true>': 'l2q_vnet_heads_vupdate_sliced_tape_f64',
    'su3_expm_mul2_bwd_kernel': 'l2q_su3_expm_mul2_bwd', 'su3_expm_mul_bwd_kernel': 'l2q_su3_expm_mul_bwd',
    'v_update_bwd_pair_cplx_kernel<true, true>': 'l2q_v_update_bwd_pair_c128',
    'v_update_bwd_cplx_kernel<true, false>': 'l2q_v_update_bwd_c128',
    'su3_projsu_vec8_bwd_kernel': 'l2q_su3_projsu_vec8_bwd',
    'su3_force_link_kernel<2': 'l2q_su3_force_bwd',
    'scaled_tanh_bwd_sums_kernel': 'l2q_scaled_tanh_bwd_sums',
    'gemm_sliced_kernel': 'l2q_gemm_sliced_f64',
    'su3_expm_mul_kernel<true>': 'l2q_su3_expm_mul2', 'su3_expm_mul_kernel<true, false>': 'l2q_su3_expm_mul2',
    'su3_project_kernel<1>': 'l2q_su3_projsu_vec8',
    # cfg-3 (tools/kprof_u1_cfg3.py)
    # (rocprofv3 leaves the _Float16 instantiations mangled)
    'u1_heads_kstream_h_kernelIDF16_Lb0': 'l2q_u1_heads_update_h:v', 'u1_heads_kstream_h_kernelIDF16_Lb1': 'l2q_u1_heads_update_h:x',
    'gemm_skinny_h_kernelIDF16_Lb0': 'l2q_gemm_h:input', 'gemm_skinny_h_kernelIDF16_Lb1': 'l2q_gemm_h_u1x',
    'u1_force_staged_f32_kernel': 'l2q_u1_force',
}


def short(n):
    return re.sub(r'\(.*', '', n).replace('void ', '')[:80]


def main():
    files = glob.glob(sys.argv[1])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    desc = os.environ.get('L2Q_KPROF_DESC', f'SU(3) {"x".join(map(str, LAT))}, {NB} chains, fp64')
    lines = [f'# rocprofv3 --pmc passes (separate runs per counter group) of {os.environ.get("L2Q_KPROF_SCRIPT", "tools/kprof.py")}:',
             f'# {desc}.  FETCH_SIZE / WRITE_SIZE in KiB as reported; on gfx950',
             '# FETCH_SIZE counts 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM',
             '# section) -> read bytes = 2 * FETCH_SIZE * 1024.  Means over the launches of a run.', '']
    traffic = {}
    for k, v in agg.items():
        if 'l2q' not in k:
            continue
        lines.append(k)
        for c, vals in sorted(v.items()):
            lines.append(f'    {c:28s} launches={len(vals):3d} mean={sum(vals) / len(vals):.5g}')
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            rd = 2 * sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE']) * 1024
            wr = sum(v['WRITE_SIZE']) / len(v['WRITE_SIZE']) * 1024
            lines.append(f'    -> HBM/fabric traffic per launch: read {rd / 1e6:.1f} MB (FETCH_SIZE x2) '
                         f'+ write {wr / 1e6:.1f} MB = {(rd + wr) / 1e6:.1f} MB')
            for frag, entry in ENTRY.items():
                if entry and frag in k:
                    key = f'{entry}@{"x".join(map(str, LAT))}x{NB}'
                    traffic[key] = {'read_bytes': rd, 'write_bytes': wr, 'total_bytes': rd + wr,
                                    'kernel': k.replace('l2q::', ''), 'lattice': LAT,
                                    'nchains': NB, 'source': f'{sys.argv[2]} ({k})'}
        if 'TCC_HIT_sum' in v:
            h, m = sum(v['TCC_HIT_sum']), sum(v['TCC_MISS_sum'])
            lines.append(f'    -> L2 hit rate {h / (h + m):.3f}')
    open(sys.argv[2], 'w').write('\n'.join(lines) + '\n')
    # one entry per (entry point, lattice, chains): passes at another shape are kept
    old = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
    old = {k: v for k, v in old.items() if '@' in k}
    old.update(traffic)
    json.dump(old, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    print('\n'.join(l for l in lines if '->' in l or (l and not l.startswith(' '))))


if __name__ == '__main__':
    main()
