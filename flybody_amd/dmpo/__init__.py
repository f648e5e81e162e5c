"""Distributional MPO in PyTorch-ROCm with on-GPU batched rollouts and replay.

Rebuilds the learning side of the reference (flybody/agents/*: Acme/TF DMPO + Ray actors + Reverb)
as: MLP policy / critic (networks.py), the decoupled MPO loss and the categorical TD loss
(losses.py), an on-GPU n-step accumulator + uniform FIFO replay (replay.py), and a learner with
one flat-buffer gradient all-reduce per step (learner.py).
"""
from .networks import DMPONetworks, make_networks          # noqa: F401
from .losses import MPOLoss, categorical_td_loss, l2_project   # noqa: F401
from .replay import NStepReplay, SampleToInsertRatio                              # noqa: F401
from .learner import DMPOConfig, DMPOLearner                 # noqa: F401
from .checkpoint import Checkpointer, Snapshotter, Counter, MetricsLogger, load_policy_snapshot   # noqa: F401
from .evaluator import evaluate                              # noqa: F401


def gemm_flop_per_step(batch: int = 256, num_samples: int = 20, nobs: int = 741, nact: int = 59) -> int:
    """Matrix-product flops (2 per multiply-add) of ONE learner update at the reference's network shapes
    (agents/network_factory.py:82-103, learning_dmpo.py:223-263): policy MLP (256, 256, 256) + two heads, critic MLP (512, 512, 256)
    + 51 atoms.  Target policy forward + online policy forward and backward (dX + dW) = 4 policy passes; the target critic on the
    N x B sampled actions, its observation half computed once per row and broadcast over the N samples; the online critic forward +
    backward = 3 passes.  This is the figure `roofline.achieved` of bench.py: dmpo_mode and tools/learner_bench.py divide by time."""
    policy = batch*(nobs*256 + 256*256*2 + 256*2*nact)
    critic_row = 512*512 + 512*256 + 256*51
    return 2*(4*policy + num_samples*batch*(nact*512 + critic_row) + batch*nobs*512 + 3*batch*((nobs + nact)*512 + critic_row))
