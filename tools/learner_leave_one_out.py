#!/usr/bin/env python3
"""What would fusing / removing a kernel of the learner step be worth AT MOST?  Runs tools/learner_bench.py with selected kernels LEFT OUT
(their outputs replaced by cached tensors: the results are WRONG, only the timing is meaningful):
    SKIP=ha            the action-half GEMM of the target critic (5120 x 59 x 512, 24 us)
    SKIP=ha,l2,l3      all three 5120-row products of the target critic (74 us)
    SKIP=adam          the optimizer launch (13.6 us)
    python tools/learner_leave_one_out.py          (SKIP in the environment; profiles/r6/learner_leave_one_out.txt)"""
import os, sys, runpy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from flybody_amd.dmpo import fused
mode = os.environ.get('SKIP', '')
_lin = fused.linear; _cache = {}
def lin(x, w, bias=None, elu=False):
    rows = x.numel()//x.shape[-1]
    key = None
    if 'ha' in mode and x.shape[-1] == 59 and rows == 5120: key = 'ha'
    if 'l2' in mode and rows == 5120 and tuple(w.shape) == (512, 512): key = 'l2'
    if 'l3' in mode and rows == 5120 and tuple(w.shape) == (256, 512): key = 'l3'
    if key:
        if key not in _cache: _cache[key] = torch.zeros(*x.shape[:-1], w.shape[0], device=x.device)
        return _cache[key]
    return _lin(x, w, bias, elu)
fused.linear = lin
import flybody_amd.dmpo.learner as lr
if 'adam' in mode: lr.DMPOLearner._apply_gradients = lambda self: None
if 'td' in mode:
    _td = fused.td_loss_grad; _tc = {}
    def td(*a, **k):
        if 'r' not in _tc: _tc['r'] = _td(*a, **k)
        return _tc['r']
    fused.td_loss_grad = td
if 'mpo' in mode:
    _mp = fused.mpo_loss_grad; _mc = {}
    def mp(*a, **k):
        if 'r' not in _mc: _mc['r'] = _mp(*a, **k)
        return _mc['r']
    fused.mpo_loss_grad = mp
sys.argv = ['learner_bench.py', '--steps', '600']
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'learner_bench.py'), run_name='__main__')
