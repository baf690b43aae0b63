"""Summarise the PMC passes of tools/prof_round.sh into profiles/<round>/traffic.json (read by bench.py for roofline.traffic).

  FETCH_SIZE / WRITE_SIZE (KB, separate passes) -> HBM bytes per launch, gfx950-corrected as MI355X_MICROARCH.md prescribes:
  FETCH_SIZE counts 16-byte-per-lane loads at half their bytes (x 2), WRITE_SIZE is exact.
  SQ_VALU_MFMA_BUSY_CYCLES (summed over SIMDs) / (1024 SIMDs x duration x 2.4 GHz) -> MFMA-busy fraction.
usage: python tools/make_traffic_json.py gpurun_out/r03_prof batch length metrics precision [round=r03]
The file records the sha256 of nele_gan_amd/csrc it was taken on; bench.py drops the counters when the kernel sources have changed since.
"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
ROUND = sys.argv[6] if len(sys.argv) > 6 else 'r05'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
work = {'batch': int(sys.argv[2]), 'length': int(sys.argv[3]), 'metrics': sys.argv[4], 'precision': sys.argv[5]}
WANT = ('glayer16_kernel<2, 8, 4, 0>', 'glayer16_kernel<2, 8, 4, 1>', 'glayer16_kernel<1, 4, 2, 0>', 'cln_bwd_kernel', 'conv_wgrad_dma_kernel<4, 11, 3>', 'conv_wgrad_dma_kernel<4, 11, 2>', 'conv_wgrad_dma_kernel<3, 13, 4>', 'conv_wgrad_dma_kernel<2, 4, 4>', 'conv_wgrad_dma_kernel<1, 1, 4>',
        'eigh_tridiag_cluster2_kernel', 'conv_wgrad_tile16_kernel<4, 7, false, false>', 'conv16_kernel<4, 4, false>', 'conv16_kernel<4, 4, true>', 'conv16_kernel<3, 4, true>', 'conv16_kernel<2, 4, true>', 'conv16_kernel<2, 8, true>', 'conv16_kernel<1, 8, true>', 'conv16_kernel<1, 8, false>', 'conv16_kernel<1, 4, true>',
        'conv_wgrad_tile16_kernel<4, 7, true, true>', 'conv_wgrad_tile16_kernel<3, 4, true, true>', 'conv_wgrad_tile16_kernel<2, 2, true, true>', 'conv_tile16_kernel<4, 8>', 'conv_tile16_kernel<3, 8>', 'conv_tile16_kernel<2, 8>', 'conv_tile16_kernel<3, 4>', 'conv_tile16_kernel<2, 4>', 'conv_tile16_kernel<1, 4>', 'conv_span16_kernel<3>', 'conv_wgrad_tile16_kernel<4, 7, true>', 'conv_wgrad_tile16_kernel<4, 7, false>', 'conv_wgrad_tile16_kernel<3, 4, false>',
        'haspi_gain_lp_sl_kernel', 'haspi_ihc_fir_kernel', 'haspi_ihc_fir9_kernel', 'haspi_mod_slide_kernel<0>', 'haspi_cep_kernel', 'haspi_resample_kernel', 'haspi_bank_tail_kernel<true>', 'haspi_bank_tail_kernel<false>',
        'siib_quad_kernel', 'siib_lag_kernel<0, 1, -14, 29>', 'eigh_invit_kernel', 'eigh_bisect_kernel', 'conv_wgrad_tile16_kernel<1, 2, true, true>', 'haspi_bank_scan_kernel<true, true, false, true>', 'haspi_bank_scan_kernel<false, true, false, false>',
        'eigh_backtransform_wy_kernel', 'eigh_tridiag_mid_kernel', 'siib_stack_kernel',
        'haspi_mod_slide_kernel<1>', 'stft_band_kernel', 'gain_istft_kernel', 'siib_proj_kernel<2>', 'siib_cov_kernel',
        'eigh_tridiag_cluster4_kernel', 'eigh_tridiag_clusters_kernel', 'eigh_invit2_kernel', 'conv1d_tile16_kernel<4>', 'adam_guarded_kernel',
        'siib_spec_wave_kernel', 'siib_spec_kernel', 'siib_db_kernel', 'stft_band_wave_kernel', 'gain_istft_wave_kernel', 'eigh_tridiag_midx_kernel', 'estoi_resample5_kernel')


def key(name):
    for w in WANT:
        if w in name:
            return w
    return None


acc = defaultdict(lambda: defaultdict(list))
for d in ('pmc_FETCH_SIZE', 'pmc_WRITE_SIZE', 'pmc_SQ'):
    for f in sorted(glob.glob('%s/%s/*/*_counter_collection.csv' % (root, d)), key=os.path.getmtime)[-1:]:       # the newest run only (merged gpurun_out directories keep old ones)
        for r in csv.DictReader(open(f)):
            k = key(r['Kernel_Name'])
            if k:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
                if d == 'pmc_SQ' and r['Counter_Name'] == 'SQ_WAVE_CYCLES':
                    acc[k]['_dur_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
out = {'note': __doc__.strip().split('\n\n')[0] + ' Averages per launch over all launches of the kernel in `bench.py --steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated`.',
       'csrc_sha': bench.csrc_sha(),
       'workload': work, 'kernels': {}}
for k, c in acc.items():
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    e = {'launches': len(c.get('FETCH_SIZE', c.get('SQ_WAVE_CYCLES', [])))}
    if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg:
        e.update({'FETCH_SIZE_KB': avg['FETCH_SIZE'], 'WRITE_SIZE_KB': avg['WRITE_SIZE'],
                  'hbm_bytes_corrected': int(avg['FETCH_SIZE'] * 1024 * 2 + avg['WRITE_SIZE'] * 1024)})
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and avg.get('_dur_ns'):
        e.update({'SQ_VALU_MFMA_BUSY_CYCLES': avg['SQ_VALU_MFMA_BUSY_CYCLES'], 'SQ_INSTS_VALU_MFMA_MOPS_BF16': avg.get('SQ_INSTS_VALU_MFMA_MOPS_BF16'),
                  'SQ_BUSY_CU_CYCLES': avg.get('SQ_BUSY_CU_CYCLES'), 'SQ_WAVE_CYCLES': avg.get('SQ_WAVE_CYCLES'), 'pmc_pass_duration_us': avg['_dur_ns'] / 1e3,
                  'mfma_busy_frac': avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * avg['_dur_ns'] * 2.4)})
    out['kernels'][k] = e
os.makedirs('profiles/%s' % ROUND, exist_ok=True)
json.dump(out, open('profiles/%s/traffic.json' % ROUND, 'w'), indent=1)
for k, e in sorted(out['kernels'].items()):
    print('%-40s %s' % (k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a in ('hbm_bytes_corrected', 'mfma_busy_frac', 'pmc_pass_duration_us', 'launches')}))
