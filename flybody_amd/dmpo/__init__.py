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
