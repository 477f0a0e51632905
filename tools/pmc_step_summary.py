"""Fold the three rocprofv3 --pmc passes of scripts/pmc_step.sh into one JSON: HBM-side bytes per denoising step (total and per
kernel) and MFMA-pipe utilisation per kernel.

    python tools/pmc_step_summary.py <dir with fetch/write/sq *_counter_collection.csv> <steps incl. warm-up> [--tag r02]

Units and corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B per lane) streaming reads at 64 B, so the read side is DOUBLED; WRITE_SIZE is taken as reported
(uncalibrated).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz) per kernel, i.e. the fraction of
the PEAK matrix-pipe cycles (the clock the 2.5 PFLOP/s figure assumes) in which a SIMD executed an MFMA (32 busy cycles per
v_mfma_f32_32x32x16_bf16); durations are the dispatch timestamps of the same pass.  (GRBM_GUI_ACTIVE is collected too but
spans the profiler's per-dispatch counter start/stop, 10x the kernel: it is reported, not used.)
Kernels that only run in the once-per-call prepare phase are listed but excluded from the per-step totals.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PREPARE_ONLY = ('k_linear_f32', 'k_mod_finalize', 'k_rope_table', 'k_headnorm', 'k_cast_bf16', 'k_conv1d', 'at::native', '__amd_rocclr')
N_SIMD = 256 * 4
PEAK_GHZ = 2.4


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:80]


def read_pass(d, tag):
    files = glob.glob(os.path.join(d, '**', f'*{tag}*counter_collection.csv'), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            did = r.get('Dispatch_Id', r.get('Correlation_Id', len(calls[k])))
            try:
                calls[k][did] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            except (KeyError, ValueError):
                calls[k][did] = 0.0
    for k, v in calls.items():
        agg[k]['_dur_ns'] = sum(v.values())
    return agg, {k: len(v) for k, v in calls.items()}, files


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('steps', type=int)
    ap.add_argument('--tag', default='r02')
    ap.add_argument('--args', default='')
    a = ap.parse_args()
    from ezaudio_amd.build import source_hash
    fetch, fcalls, ff = read_pass(a.dir, 'fetch')
    write, _, wf = read_pass(a.dir, 'write')
    sq, scalls, sf = read_pass(a.dir, 'sq')
    lds, _, lf = read_pass(a.dir, 'lds')     # optional passes: absent in summaries made before round 3
    tcc, _, tf = read_pass(a.dir, 'tcc')
    dram, _, df = read_pass(a.dir, 'dram')
    kernels = sorted(set(fetch) | set(write) | set(sq))
    rows, tot_f, tot_w, tot_busy, tot_gui = [], 0.0, 0.0, 0.0, 0.0
    for k in kernels:
        prep = any(p in k for p in PREPARE_ONLY)
        fb = 2.0 * fetch.get(k, {}).get('FETCH_SIZE', 0.0) * 1024.0      # x2: gfx950 correction for 16 B / lane streaming reads
        wb = write.get(k, {}).get('WRITE_SIZE', 0.0) * 1024.0
        s = sq.get(k, {})
        gui = s.get('GRBM_GUI_ACTIVE', 0.0)
        busy = s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        dur = s.get('_dur_ns', 0.0)
        n = fcalls.get(k, scalls.get(k, 0))
        row = {'kernel': k, 'dispatches': n, 'prepare_only': prep, 'fetch_bytes': fb, 'write_bytes': wb,
               'fetch_bytes_per_dispatch': fb / n if n else 0.0, 'write_bytes_per_dispatch': wb / n if n else 0.0,
               'mfma_busy_frac': busy / (N_SIMD * dur * PEAK_GHZ) if dur else 0.0, 'mfma_insts': s.get('SQ_INSTS_MFMA', 0.0),
               'duration_us_per_dispatch': dur / n / 1e3 if n else 0.0,
               'wave_cycles': s.get('SQ_WAVE_CYCLES', 0.0), 'wait_any': s.get('SQ_WAIT_ANY', 0.0),
               'wait_inst_any': s.get('SQ_WAIT_INST_ANY', 0.0), 'active_inst_any': s.get('SQ_ACTIVE_INST_ANY', 0.0),
               'gui_active_cycles': gui}
        if k in lds:   # LDS pipe: conflict cycles per active LDS cycle, LDS instructions, and the share of wave time spent waiting on LDS
            q = lds[k]
            row.update({'lds_bank_conflict_cycles': q.get('SQ_LDS_BANK_CONFLICT', 0.0), 'lds_idx_active_cycles': q.get('SQ_LDS_IDX_ACTIVE', 0.0),
                        'lds_bank_conflict_frac': q.get('SQ_LDS_BANK_CONFLICT', 0.0) / q['SQ_LDS_IDX_ACTIVE'] if q.get('SQ_LDS_IDX_ACTIVE') else 0.0,
                        'lds_addr_conflict_cycles': q.get('SQ_LDS_ADDR_CONFLICT', 0.0), 'lds_insts': q.get('SQ_INSTS_LDS', 0.0),
                        'valu_insts': q.get('SQ_INSTS_VALU', 0.0),
                        'wait_inst_lds_frac_of_wave_cycles': q.get('SQ_WAIT_INST_LDS', 0.0) / q['SQ_WAVE_CYCLES'] if q.get('SQ_WAVE_CYCLES') else 0.0,
                        'active_inst_lds_frac_of_wave_cycles': q.get('SQ_ACTIVE_INST_LDS', 0.0) / q['SQ_WAVE_CYCLES'] if q.get('SQ_WAVE_CYCLES') else 0.0})
        if k in tcc:   # L2 (TCC, summed over the 8 XCDs x 16 channels): hit rate of the requests the CUs sent, and reads forwarded to the fabric
            q = tcc[k]
            hm = q.get('TCC_HIT_sum', 0.0) + q.get('TCC_MISS_sum', 0.0)
            row.update({'tcc_hit': q.get('TCC_HIT_sum', 0.0), 'tcc_miss': q.get('TCC_MISS_sum', 0.0), 'tcc_req': q.get('TCC_REQ_sum', 0.0),
                        'tcc_hit_rate': q.get('TCC_HIT_sum', 0.0) / hm if hm else 0.0, 'tcc_ea_rdreq': q.get('TCC_EA0_RDREQ_sum', 0.0)})
        if k in dram:  # requests the L2s sent to the fabric that were routed to DRAM (the others hit the Infinity Cache); request counts, not bytes
            q = dram[k]
            row.update({'ea_rdreq_dram': q.get('TCC_EA0_RDREQ_DRAM_sum', 0.0), 'ea_wrreq_dram': q.get('TCC_EA0_WRREQ_DRAM_sum', 0.0),
                        'ea_wrreq': q.get('TCC_EA0_WRREQ_sum', 0.0), 'tcc_bubble': q.get('TCC_BUBBLE_sum', 0.0)})
            if k in tcc and tcc[k].get('TCC_EA0_RDREQ_sum'):
                row['ea_rdreq_dram_frac'] = q.get('TCC_EA0_RDREQ_DRAM_sum', 0.0) / tcc[k]['TCC_EA0_RDREQ_sum']
        rows.append(row)
        if not prep:
            tot_f += fb; tot_w += wb; tot_busy += busy; tot_gui += dur
    rows.sort(key=lambda r: -(r['fetch_bytes'] + r['write_bytes']))
    out = {'tag': a.tag, 'src_hash': source_hash(), 'steps_profiled': a.steps, 'bench_args': a.args,
           'fetch_bytes_per_step': tot_f / a.steps, 'write_bytes_per_step': tot_w / a.steps,
           'traffic_bytes_per_step': (tot_f + tot_w) / a.steps,
           'mfma_busy_frac': tot_busy / (N_SIMD * tot_gui * PEAK_GHZ) if tot_gui else 0.0,
           'note': 'FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported; prepare-only kernels excluded; '
                   'eager launches (--no-graph) so each dispatch is attributed; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel ns x 2.4 GHz)',
           'files': [os.path.basename(f) for f in ff + wf + sf + lf + tf + df], 'kernels': rows}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
