"""Posterior post-processing (probdiffeq.stats surface used by the reference).

src/odecheckpts/ivpsolvers.py:80-81 computes
    markov_seq = stats.markov_select_terminal(sol.posterior)
    margs      = stats.markov_marginals(markov_seq, reverse=True)
i.e. the backward pass over the stored transitions; :118,144 evaluates the posterior between the stored points
(`offgrid_marginals_searchsorted`), experiments/5_vs_interpolation/measure.py:69-75 draws joint samples (`markov_sample`).
On MI355X all three run on the device (`pnode_smooth_checkpoint_means`, `pnode_offgrid_means`, `pnode_sample_checkpoints`)
against the store a fixed-point / smoother solve left there; the functions here expose them with the reference's call shapes.
"""

from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass
class MarkovSeq:
    means: np.ndarray  # (B, T, n, d) smoothed means at all stored time points
    terminal: np.ndarray  # (B, n, d)
    posterior: object = None


@dataclasses.dataclass
class Marginals:
    mean: np.ndarray


def markov_select_terminal(posterior) -> MarkovSeq:
    if posterior.kind not in ("fixedpoint", "smoother"):
        raise ValueError("markov_select_terminal needs a posterior with a backward model (fixedpoint / smoother)")
    return MarkovSeq(posterior.means, posterior.means[:, -1], posterior)


def markov_marginals(markov_seq: MarkovSeq, *, reverse: bool) -> Marginals:
    """Marginals at all stored points but the terminal one (`markov_seq.terminal`), ordered in time."""
    if not reverse:
        raise ValueError("only reverse=True (backward marginalisation from the terminal state) is implemented")
    return Marginals(markov_seq.means[:, :-1])


def _store(posterior):
    if posterior is None or posterior.batch is None or posterior.grid is None:
        raise ValueError("the posterior carries no stored backward model (use strategy_fixedpoint with solve_adaptive_save_at "
                         "or strategy_smoother with solve_adaptive_save_every_step)")
    return posterior.batch, posterior.grid, np.asarray(posterior.num_filled, dtype=np.int32)


def offgrid_marginals_searchsorted(*, ts, solution, solver=None):
    """Posterior marginals at the times `ts` (all inside the solution's time span), between the stored points included:
    `dense, marginals = offgrid_marginals_searchsorted(ts=save_at, solution=sol, solver=solver)` (ivpsolvers.py:118,144).
    Returns the zeroth-derivative means (len(ts), d) and Marginals(mean (len(ts), n, d)); with a leading batch axis when the
    solution describes a whole batch."""
    post = solution.posterior
    batch, grid, nf = _store(post)
    ts = np.atleast_1d(np.asarray(ts, dtype=np.float64))
    B, Q = batch.B, ts.shape[0]
    t = np.broadcast_to(ts, (B, Q)).copy()
    t_lo, t_hi = np.zeros((B, Q)), np.ones((B, Q))
    idx = np.full((B, Q), -1, dtype=np.int32)
    at_grid = np.zeros((B, Q), dtype=np.uint8)
    at_t0 = np.zeros((B, Q), dtype=bool)
    which = range(B) if post.index is None else [post.index]
    for b in which:
        g = grid[b, : nf[b] + 1]
        if np.any(ts < g[0]) or np.any(ts > g[-1]):
            raise ValueError(f"ts must lie inside the solution's time span [{g[0]}, {g[-1]}]")
        i = np.searchsorted(g[1:], ts, side="left")  # the interval (g[i], g[i+1]] holds ts
        at_t0[b] = ts == g[0]
        idx[b] = np.where(at_t0[b], -1, i)
        t_lo[b], t_hi[b] = g[i], g[i + 1]
        at_grid[b] = ts == g[i + 1]
    means = batch.offgrid_means(nf, post.init_means, t, t_lo, t_hi, idx, at_grid)
    if at_t0.any():  # the smoothed marginal at the initial time is part of the backward pass
        sm0 = batch.smooth_checkpoint_means(nf)[:, 0]
        for b in which:
            means[b, at_t0[b]] = sm0[b]
    if post.index is not None:
        means = means[post.index]
    return means[..., 0, :], Marginals(means)


def markov_sample(key, markov_seq: MarkovSeq, *, shape=(), reverse: bool = True):
    """Joint samples of the posterior at its stored time points, drawn backwards from the terminal marginal:
    `(qoi, samples), (init_qoi, init) = markov_sample(key, markov_seq, shape=(S,), reverse=True)` (measure.py:69-75).
    `qoi` (S, T-1, d) / `samples` (S, T-1, n, d) are ordered in time WITHOUT the terminal point, `init_*` is the sample of the
    terminal marginal (the "init" of the backward sequence), as in the reference. `key`: an int seed or a numpy Generator (the
    base samples are drawn on the host, the recursion runs on the device). A whole batch gets a leading axis."""
    if not reverse:
        raise ValueError("only reverse=True is implemented")
    post = markov_seq.posterior
    batch, grid, nf = _store(post)
    shape = tuple(int(s) for s in np.atleast_1d(shape)) if np.size(shape) else ()
    S = int(np.prod(shape)) if shape else 1
    rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
    # (sized by the entries the solve filled, not by the store's capacity: a store-every-step smoother holds up to 16 384)
    eps = rng.standard_normal((batch.B, S, int(np.max(nf)) + 1, batch.n, batch.d))
    out = batch.sample_checkpoints(nf, eps)
    which = list(range(batch.B)) if post.index is None else [post.index]
    if len({int(nf[b]) for b in which}) != 1:
        raise ValueError("markov_sample on a whole batch needs the same number of stored points in every trajectory")
    T1 = int(nf[which[0]])
    x = out[which][:, :, : T1 + 1]  # (b, S, T, n, d)
    x = x.reshape((len(which),) + (shape if shape else ()) + x.shape[2:]) if shape else x[:, 0]
    if post.index is not None:
        x = x[0]
    body, init = x[..., :T1, :, :], x[..., T1, :, :]
    return (body[..., 0, :], body), (init[..., 0, :], init)
