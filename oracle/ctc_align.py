"""CPU restatement of the reference's only native kernel, CTC best alignment -- TEST INFRASTRUCTURE ONLY
(SURVEY 8(f) row 4).

`alignment_kernel` follows criterion/best_alignment/best_alignment.cu:58-170 (ctc_alignment_log_alpha_gpu_kernel: the
Viterbi recurrence over the 2T+1 CTC states, predecessor preference s, s-1, s-2 under strict '>' comparisons);
`best_alignment` follows criterion/best_alignment/__init__.py:25-111 (final state among the last two reachable states,
back-tracking, optional state -> label translation).  The .cu file needs nvcc / ATen CUDA headers: unbuildable here.
Pinned two ways: tests/golden/g16_best_alignment.npz records the reference's own Python wrapper running over this
kernel restatement, and tests compare the result with a brute-force maximum over ALL alignments on tiny cases."""
import numpy as np

NEG_INF = -np.inf


def target_prime(targets_row, idx, blank):
    return blank if idx % 2 == 0 else int(targets_row[idx // 2])


def alignment_kernel(log_probs, targets, input_lengths, target_lengths, blank):
    """log_probs [S, N, V] float; targets [N, Tmax] int.  Returns (neg_log_likelihood [N], log_alpha [N, S, 2Tmax+1],
    paths [N, S, 2Tmax+1] int64, -1 where never written)."""
    S, N, V = log_probs.shape
    max_t = int(max(target_lengths)) if N else 0
    ns = 2 * max_t + 1
    la = np.full((N, S, ns), NEG_INF, dtype=log_probs.dtype)
    paths = np.full((N, S, ns), -1, dtype=np.int64)
    nll = np.zeros(N, dtype=log_probs.dtype)
    for b in range(N):
        L, T = int(input_lengths[b]), int(target_lengths[b])
        la[b, 0, 0] = log_probs[0, b, blank]                                   # .cu:83-104
        if ns > 1:
            la[b, 0, 1] = NEG_INF if T == 0 else log_probs[0, b, target_prime(targets[b], 1, blank)]
        for s in range(ns):
            if s < 2 * T + 1 and T > 0:                                        # :111-122
                cur = target_prime(targets[b], s, blank)
                three = s > 1 and target_prime(targets[b], s - 2, blank) != cur
            else:
                cur, three = blank, False
            for t in range(1, S):                                              # :123-168
                if t < L and s < 2 * T + 1:
                    lamax, mp = la[b, t - 1, s], s
                    if s > 0 and la[b, t - 1, s - 1] > lamax:
                        lamax, mp = la[b, t - 1, s - 1], s - 1
                    if three and la[b, t - 1, s - 2] > lamax:
                        lamax, mp = la[b, t - 1, s - 2], s - 2
                    la[b, t, s] = lamax + log_probs[t, b, cur]
                    paths[b, t, s] = mp
        l1 = la[b, L - 1, 2 * T]                                               # :172-186
        l2 = la[b, L - 1, 2 * T - 1] if T > 0 else NEG_INF
        m = max(l1, l2)
        m = 0.0 if m == NEG_INF else m
        with np.errstate(divide="ignore"):
            nll[b] = -(np.log(np.exp(l1 - m) + np.exp(l2 - m)) + m)
    return nll, la, paths


def best_alignment(log_prob, targets, input_lengths, target_lengths, blank=0, as_labels=False):
    """__init__.py:25-111.  Returns [N, S] int64 states (or labels)."""
    log_prob, targets = np.asarray(log_prob), np.asarray(targets)
    input_lengths, target_lengths = np.asarray(input_lengths), np.asarray(target_lengths)
    _, la, paths = alignment_kernel(log_prob, targets, input_lengths, target_lengths, blank)
    N, S, ns = la.shape
    out = np.zeros((N, S), dtype=np.int64)
    for b in range(N):
        L, sl = int(input_lengths[b]), 2 * int(target_lengths[b]) + 1
        col = la[b, L - 1]                                                     # :63-76
        isinf = np.isneginf(col)
        first = int(np.argmax(isinf)) if isinf.any() else 0
        last = min((first - 1) % sl, sl - 2)                                   # :82 (python % like torch.remainder)
        masked = la[b].copy()                                                  # :83-85
        keep = (np.arange(ns) >= last) & (np.arange(ns) < sl)
        masked[:, ~keep] = NEG_INF
        dec = masked.argmax(axis=1)                                            # :89 (first max; all -inf -> 0)
        for t in range(S - 1, 0, -1):                                          # :91-97
            if t < L:
                dec[t - 1] = paths[b, t, dec[t]]
        out[b] = dec
    if as_labels:                                                              # :101-107
        lab = np.where(out % 2 == 1, np.take_along_axis(targets, np.minimum(out // 2, targets.shape[1] - 1), axis=1), blank)
        return lab.astype(np.int64)
    return out


def brute_force_best_state_path(log_probs_b, target, blank):
    """max over ALL monotone CTC state paths of length S ending in one of the last two states (tiny cases only).
    Returns (best score, list of all state paths attaining it)."""
    S = log_probs_b.shape[0]
    T = len(target)
    ns = 2 * T + 1
    lab = [blank if s % 2 == 0 else int(target[s // 2]) for s in range(ns)]
    best, arg = NEG_INF, []

    def rec(t, s, score, path):
        nonlocal best, arg
        score = score + log_probs_b[t, lab[s]]
        path = path + [s]
        if t == S - 1:
            if s >= ns - 2:
                if score > best + 1e-12:
                    best, arg = score, [path]
                elif abs(score - best) <= 1e-12:
                    arg.append(path)
            return
        for ns_ in (s, s + 1, s + 2):
            if ns_ >= ns:
                continue
            if ns_ == s + 2 and not (lab[ns_] != blank and lab[ns_] != lab[s]):
                continue
            rec(t + 1, ns_, score, path)

    for s0 in (0, 1):
        if s0 < ns:
            rec(0, s0, 0.0, [])
    return best, arg
