"""Posterior post-processing (probdiffeq.stats surface used by the reference).

src/odecheckpts/ivpsolvers.py:80-81 computes
    markov_seq = stats.markov_select_terminal(sol.posterior)
    margs      = stats.markov_marginals(markov_seq, reverse=True)
i.e. the backward pass over the len(save_at)-1 stored checkpoint transitions. On MI355X that pass runs
on the device (`pnode_smooth_checkpoint_means`) at the end of `solve_adaptive_save_at`; the functions
here expose its result with the reference's call shapes.
"""

from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass
class MarkovSeq:
    means: np.ndarray  # (B, T, n, d) smoothed means at all save_at points
    terminal: np.ndarray  # (B, n, d)


@dataclasses.dataclass
class Marginals:
    mean: np.ndarray


def markov_select_terminal(posterior) -> MarkovSeq:
    if posterior.kind != "fixedpoint":
        raise ValueError("markov_select_terminal needs a fixed-point-smoother posterior")
    return MarkovSeq(posterior.means, posterior.means[:, -1])


def markov_marginals(markov_seq: MarkovSeq, *, reverse: bool) -> Marginals:
    """Marginals at save_at[:-1] (the terminal one is `markov_seq.terminal`), ordered like save_at."""
    if not reverse:
        raise ValueError("only reverse=True (backward marginalisation from the terminal state) is implemented")
    return Marginals(markov_seq.means[:, :-1])
