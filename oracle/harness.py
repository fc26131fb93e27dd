"""CPU restatement of the step AFTER the hot path -- TEST INFRASTRUCTURE ONLY.

`units_to_segment` follows agents/default_agent.py:248-301 (subword queue -> words for the SimulEval server) and is
pinned to tests/golden/g15_units_to_segment.json, recorded from the reference method itself.  The latency scorers
restate SimulEval's published definitions (SimulEval absent: parity unpinned), `*_CA` = the same formulas over
computation-aware delays (source time at commit + wall-clock spent computing so far)."""
BOW_PREFIX = "▁"
DEFAULT_EOS = "</s>"


class Queue:
    """simuleval ListEntry as the reference uses it: FIFO over .value, pop() removes the first element."""

    def __init__(self, value=None):
        self.value = list(value or [])

    def __len__(self):
        return len(self.value)

    def __getitem__(self, i):
        return self.value[i]

    def append(self, v):
        self.value.append(v)

    def pop(self, index=0):
        return self.value.pop(index)


def dict_string(symbols, eos, tokens, bpe_symbol=None):
    s = " ".join(symbols[int(t)] for t in tokens if int(t) != eos)
    if bpe_symbol == "sentencepiece":
        s = s.replace(" ", "").replace(BOW_PREFIX, " ").strip()
    return s


def units_to_segment(unit_queue, symbols, eos, n_target, max_len):
    """:248-301.  n_target = len(states.units.target); max_len = self.max_len(len(states.units.source))."""
    if eos == unit_queue[0]:                                   # :262-263
        return DEFAULT_EOS
    segment = []
    if None in unit_queue.value:                               # :267-268 (force finish)
        unit_queue.value.remove(None)
    if (len(unit_queue) > 0 and eos == unit_queue[-1]) or n_target > max_len:    # :271-281
        return [dict_string(symbols, eos, unit_queue.value, "sentencepiece")] + [DEFAULT_EOS]
    for index in list(unit_queue.value):                       # :283-299
        token = dict_string(symbols, eos, [index])
        if token.startswith(BOW_PREFIX):
            if len(segment) == 0:
                segment += [token.replace(BOW_PREFIX, "")]
            else:
                for _ in range(len(segment)):
                    unit_queue.pop()
                out = ["".join(segment)]
                if eos == unit_queue[0]:
                    out += [DEFAULT_EOS]
                return out
        else:
            segment += [token.replace(BOW_PREFIX, "")]
    return None


def latency_scores(delays, elapsed, src_len_ms):
    """AL / AP / DAL and their computation-aware twins from per-unit delays (ms of source consumed at commit) and
    elapsed (same + compute wall-clock ms).  Formulas: oracle.latency."""
    from .latency import average_lagging, average_proportion, differentiable_average_lagging
    out = {}
    for tag, d in (("", delays), ("_CA", elapsed)):
        out["AL" + tag] = average_lagging(d, src_len_ms)
        out["AP" + tag] = average_proportion(d, src_len_ms)
        out["DAL" + tag] = differentiable_average_lagging(d, src_len_ms)
    return out
