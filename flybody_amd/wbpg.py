"""Wing-beat pattern generator tables.

Restates the table construction of flybody/tasks/pattern_generators.py:17-129: 201 wing-angle sequences for
beat frequencies 218 Hz +/- 5 %, each a whole number of base-pattern repeats resampled at the control timestep
with the repeat count chosen for the smoothest wrap-around.  The per-environment state machine
(step / frequency index / filtered frequency, pattern_generators.py:131-203) runs inside the step kernel; this
module only builds the flat tables it consumes.
"""
from __future__ import annotations

import numpy as np

BASE_FREQ = 218.0          # tasks/constants.py:24
REL_FREQ_RANGE = 0.05
NUM_FREQS = 201


def default_base_pattern() -> np.ndarray:
    x = np.linspace(0, 2*np.pi, 500)
    yaw = 1.1*np.sin(x - np.pi/2) + 0.3
    roll = 0.25*np.sin(1.5*x) - 0.1
    pitch = 1.35*np.sin(x) + 0.8
    return np.vstack((yaw, roll, pitch)).T


def build_tables(base_pattern: np.ndarray | None = None, base_beat_freq: float = BASE_FREQ, rel_freq_range: float = REL_FREQ_RANGE,
                 num_freqs: int = NUM_FREQS, min_repeats: int = 10, max_repeats: int = 20, dt_ctrl: float = 2e-4,
                 ctrl_filter: float | None = None):
    if base_pattern is None:
        base_pattern = default_base_pattern()
    if ctrl_filter is None:
        ctrl_filter = 0.5 / base_beat_freq
    base_pattern = np.tile(base_pattern, (1, 2))
    beat_freqs = np.linspace((1 - rel_freq_range)*base_beat_freq, (1 + rel_freq_range)*base_beat_freq, num_freqs)
    trajs, phases, offs = [], [], [0]
    for beat_freq in beat_freqs:
        beat_time = 1/beat_freq
        reps = np.arange(min_repeats, max_repeats + 1)
        rel_error = ((reps*beat_time) % dt_ctrl)/dt_ctrl
        a1 = np.argmin(rel_error); a2 = np.argmin(np.abs(1 - rel_error))
        if rel_error[a1] < np.abs(1 - rel_error[a2]):
            argmin, shift = a1, dt_ctrl
        else:
            argmin, shift = a2, 0.0
        n_reps = argmin + 1
        repeated = np.tile(base_pattern, reps=(n_reps, 1))
        phase = np.linspace(0, n_reps, n_reps*base_pattern.shape[0], endpoint=False)
        dt_data = beat_time/base_pattern.shape[0]
        duration = repeated.shape[0]*dt_data
        t_data = np.linspace(0, duration, repeated.shape[0])
        t_ctrl = np.arange(0, duration - shift, dt_ctrl)
        traj = np.stack([np.interp(t_ctrl, t_data, repeated[:, i]) for i in range(base_pattern.shape[1])], axis=1)
        trajs.append(traj); phases.append(np.interp(t_ctrl, t_data, phase)); offs.append(offs[-1] + len(t_ctrl))
    return dict(traj=np.concatenate(trajs), phase=np.concatenate(phases), offset=np.array(offs, np.int32), beat_freqs=beat_freqs,
                base_freq=float(base_beat_freq), rel_range=float(rel_freq_range),
                rate=float(np.exp(-dt_ctrl/ctrl_filter)) if ctrl_filter != 0 else 0.0, dt_ctrl=float(dt_ctrl))


class HostWBPG:
    """Scalar reference of the state machine (tests compare the kernel / oracle against it)."""

    def __init__(self, tables):
        self.t = tables

    def _seq(self, k):
        o = self.t['offset']
        return self.t['traj'][o[k]:o[k+1]], self.t['phase'][o[k]:o[k+1]]

    def reset(self, initial_phase=0.0):
        self.ctrl_freq = self.t['base_freq']
        self.freq_idx = int(np.argmin(np.abs(self.t['beat_freqs'] - self.ctrl_freq)))
        traj, phase = self._seq(self.freq_idx)
        self.step_i = int(np.argmin(np.abs(initial_phase - phase)))
        return traj[self.step_i], (traj[self.step_i + 1] - traj[self.step_i])/self.t['dt_ctrl']

    def step(self, ctrl_freq):
        traj, phase = self._seq(self.freq_idx)
        self.step_i = (self.step_i + 1) % len(traj)
        r = self.t['rate']
        self.ctrl_freq = ctrl_freq if r == 0 else self.ctrl_freq*r + ctrl_freq*(1 - r)
        idx_new = int(np.argmin(np.abs(self.t['beat_freqs'] - self.ctrl_freq)))
        if idx_new != self.freq_idx:
            cur = phase[self.step_i]
            ntraj, nphase = self._seq(idx_new)
            self.step_i = int(np.argmin(np.abs(cur % 1 - nphase % 1)))
            self.freq_idx = idx_new; traj = ntraj
        return traj[self.step_i]
