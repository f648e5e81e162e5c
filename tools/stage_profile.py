#!/usr/bin/env python3
"""Per-STAGE hardware counters of the step kernel: which stage issues the instructions, how many lanes are active,
which stage moves the bytes.

Two halves:
  run      -- the workload.  Steps a batch a few control steps with the fused kernel (untimed), then walks S control steps ONE STAGE PER
              LAUNCH through fb_batch_stage (fb_step.hpp MODE_STAGE: same stage functions, LDS pool parked between launches; bit-identical to
              the fused step, tests/test_kernel_emulation.py::test_single_stage_launches_equal_fused_step) and writes the launch -> stage map.
              Run it under `rocprofv3 --pmc ... --kernel-trace` (tools/collect_stage_profile.sh does, one counter group per pass).
  report   -- joins the per-dispatch counter CSVs of the passes with the stage map and prints, per stage: launches per substep, VALU / SALU /
              LDS / VMEM instructions per environment-substep, mean active lanes (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU),
              share of FP64 arithmetic instructions, wave-cycles and wait share, bytes fetched / written, and the ISA-level scratch
              instruction count of the stage function (tools/resource_report.py).

  python tools/stage_profile.py run OUT_DIR [--envs 4096] [--steps 2] [--dense]
  python tools/stage_profile.py report OUT_DIR [--dense] > profiles/r4/stage_lanes.txt
"""
import argparse, collections, csv, glob, json, os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def run(a):
    import numpy as np, torch
    from flybody_amd import engine
    from flybody_amd.reference import default_walking_reference
    M = engine.Model.from_asset('walk_imitation', dense=a.dense)
    B = engine.Batch(M, a.envs, precision=64)
    qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    act = torch.empty(a.envs, M.dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
    nfused = 0                                              # fused k_fly launches before the single-stage ones (the reset runs under its own kernel name, k_fly_reset)
    for t in range(a.warm):
        B.random_actions(act.data_ptr(), t, seed=1); B.step_ptr(act.data_ptr(), st); nfused += 1
    torch.cuda.synchronize()
    seq = engine.stage_sequence(M.dim('nsubstep'))
    names = []
    for t in range(a.steps):
        B.random_actions(act.data_ptr(), a.warm + t, seed=1)
        for name, word in seq:
            B.stage(word, act.data_ptr(), st); names.append(name)
    torch.cuda.synchronize()
    ok = bool(np.isfinite(B.get('QPOS')).all())
    os.makedirs(a.out, exist_ok=True)
    json.dump({'n_env': a.envs, 'nsubstep': M.dim('nsubstep'), 'steps': a.steps, 'fused_launches_before': nfused, 'stages': names, 'finite': ok,
               'nefc_mean': float(B.get('NEFC').mean()), 'ncon_mean': float(B.get('NCON').mean()), 'build': 'dense' if a.dense else 'default',
               'version': engine.version(engine.HIP_LIB_DENSE if a.dense else None)}, open(os.path.join(a.out, 'stage_map.json'), 'w'))
    print('stage walk ok:', len(names), 'single-stage launches, state finite', ok)


def report(a):
    mp = json.load(open(os.path.join(a.out, 'stage_map.json')))
    names, nskip, n_env, nsub, steps = mp['stages'], mp['fused_launches_before'], mp['n_env'], mp['nsubstep'], mp['steps']
    per = collections.defaultdict(lambda: collections.defaultdict(float))         # stage -> counter -> sum over launches
    count = collections.Counter(names)
    for f in sorted(glob.glob(os.path.join(a.out, 'pass_*', '**', '*counter_collection.csv'), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if 'k_fly<double>' in r['Kernel_Name'] or 'k_flyId' in r['Kernel_Name']]
        ids = sorted({int(r['Dispatch_Id']) for r in rows})
        if len(ids) != nskip + len(names):
            print(f'# {f}: {len(ids)} k_fly dispatches, expected {nskip + len(names)} -- skipped'); continue
        which = {d: names[i - nskip] for i, d in enumerate(ids) if i >= nskip}
        for r in rows:
            d = int(r['Dispatch_Id'])
            if d in which: per[which[d]][r['Counter_Name']] += float(r['Counter_Value'])
    # static scratch instruction counts of the stage functions
    scr = {}
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'resource_report.py'), 'd'] + (['-DFB_F64_DENSE=1'] if mp['build'] == 'dense' else []),
                             capture_output=True, text=True).stdout
        for l in out.splitlines()[1:]:
            p = l.split(); scr[p[0]] = (int(p[1]), int(p[2]), int(p[4]) + int(p[5]))
    except Exception as e:                                                         # noqa
        print('# resource_report failed:', e)
    fn_of = {'actuation': 's_actuation', 'factor_M': 'd_factor+d_factor_tail', 'factor_M_hD': 'd_factor+d_factor_tail', 'solve_smooth': 'd_solve', 'solve_constraint': 'd_solve',
             'solve_euler': 'd_solve', 'project_constraint': 's_project_constraint', 'constraint_solve': 's_constraint_a', 'qacc': 's_constraint_b',
             'sensor_acc': 's_sensor_acc', 'integrate': 's_integrate', 'kinematics': 's_kinematics', 'com_pos': 's_com_pos', 'crb': 's_crb',
             'collision': 's_collision+box_filter+narrow_phase', 'make_constraint': 's_make_constraint', 'velocity': 's_velocity', 'task_pre': 's_pre', 'task_post': 's_post'}
    envsub = float(n_env*nsub*steps)
    order = []
    for n in names:
        if n not in order: order.append(n)
    base = per.get('substep_end', {})                       # (nearly) empty stage: launch + LDS pool load / store only -- the overhead every row carries
    nb = count.get('substep_end', 1)
    print(f'# tools/stage_profile.py: {mp["build"]} FP64 build ({mp["version"]}), {n_env} walk_imitation environments, {steps} control steps walked one stage per launch')
    print(f'# (fb_batch_stage), mean nefc {mp["nefc_mean"]:.1f}, ncon {mp["ncon_mean"]:.1f}.  Per ENVIRONMENT-SUBSTEP (task_pre / task_post: per control step / {nsub}); every row has the')
    print('# single-stage launch overhead (LDS pool load + store = the substep_end row, itself shown raw) SUBTRACTED.  lanes = mean active lanes of a VALU')
    print('# instruction BY EXEC MASK (wave-uniform solver algebra runs with all lanes enabled although only nefc rows carry data);')
    print('# f64% = FP64 add/mul/fma/transcendental share of the VALU instructions; int% = INT32 share (address arithmetic, unpacking);')
    print('# wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES of the')
    print('# stage launch (cold L2: every stage streams the whole batch, so waits and bytes are UPPER bounds of what the fused kernel sees);')
    print('# KB rd/wr = 2 x FETCH_SIZE / WRITE_SIZE (calibration: profiles/r3/pmc_calibration.json); scr = scratch instructions in the stage function(s) (static).')
    hdr = '%-20s %5s %8s %8s %7s %7s %6s %5s %5s %9s %6s %7s %7s %5s %6s' % ('stage', 'n/sub', 'VALU', 'SALU', 'LDS', 'VMEM', 'lanes', 'f64%', 'int%', 'wavecyc', 'wait%', 'KB_rd', 'KB_wr', 'scr', 'branch')
    print(hdr)
    tot = collections.defaultdict(float)
    for n in order:
        c = per[n]; k = count[n]
        def g(name, sub=True):
            v = c.get(name, 0.0)
            if sub and n != 'substep_end': v -= base.get(name, 0.0)*k/nb
            return v/envsub
        valu, salu, lds = g('SQ_INSTS_VALU'), g('SQ_INSTS_SALU'), g('SQ_INSTS_LDS')
        vmem = g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')
        act_, thr = c.get('SQ_ACTIVE_INST_VALU', 0.0) - (0 if n == 'substep_end' else base.get('SQ_ACTIVE_INST_VALU', 0.0)*k/nb), c.get('SQ_THREAD_CYCLES_VALU', 0.0) - (0 if n == 'substep_end' else base.get('SQ_THREAD_CYCLES_VALU', 0.0)*k/nb)
        lanes = thr/act_ if act_ > 0 and thr > 0 else float('nan')      # (rocprofv3's VALUThreadUtilization: THREAD_CYCLES / (ACTIVE_INST x 64))
        f64 = sum(g(x) for x in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_TRANS_F64'))
        wc = g('SQ_WAVE_CYCLES')*4; wt = c.get('SQ_WAIT_ANY', 0.0)/max(c.get('SQ_WAVE_CYCLES', 1.0), 1.0)
        rd, wr = 2*g('FETCH_SIZE'), g('WRITE_SIZE')
        fns = fn_of.get(n, ''); sc = sum(scr.get(f, (0, 0, 0))[2] for f in fns.split('+')) if fns else 0
        i32 = g('SQ_INSTS_VALU_INT32')
        br = g('SQ_INSTS_BRANCH')          # (branch instructions, taken or not; in the insts pass since round 6)
        print('%-20s %5.1f %8.0f %8.0f %7.0f %7.0f %6.1f %5.1f %5.1f %9.0f %6.1f %7.2f %7.2f %5d %6.0f' % (n, k/float(nsub*steps), valu, salu, lds, vmem, lanes, 100*f64/valu if valu > 50 else float('nan'),
              100*i32/valu if valu > 50 else float('nan'), wc, 100*wt, rd, wr, sc, br))
        if n != 'substep_end': tot['br'] += br
        tot['i32'] += 0 if n == 'substep_end' else i32
        if n != 'substep_end':
            for key, v in (('VALU', valu), ('SALU', salu), ('LDS', lds), ('VMEM', vmem), ('wc', wc), ('rd', rd), ('wr', wr), ('f64', f64), ('thr', thr/envsub), ('act', act_/envsub)):
                tot[key] += v
    print('%-20s %5s %8.0f %8.0f %7.0f %7.0f %6.1f %5.1f %5.1f %9.0f %6s %7.2f %7.2f' % ('SUM (per env-substep)', '', tot['VALU'], tot['SALU'], tot['LDS'], tot['VMEM'],
          tot['thr']/max(tot['act'], 1e-9), 100*tot['f64']/max(tot['VALU'], 1e-9), 100*tot['i32']/max(tot['VALU'], 1e-9), tot['wc'], '', tot['rd'], tot['wr']))
    print('# x %d substeps = per env-step: VALU %.0f  SALU %.0f  LDS %.0f  VMEM %.0f  branches %.0f ; KB fetched %.1f written %.1f' % (nsub, nsub*tot['VALU'], nsub*tot['SALU'], nsub*tot['LDS'], nsub*tot['VMEM'], nsub*tot['br'], nsub*tot['rd'], nsub*tot['wr']))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('cmd', choices=['run', 'report']); ap.add_argument('out')
    ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--steps', type=int, default=2); ap.add_argument('--warm', type=int, default=8)
    ap.add_argument('--dense', action='store_true')
    a = ap.parse_args()
    run(a) if a.cmd == 'run' else report(a)
