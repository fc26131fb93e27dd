"""The step after the hot path: SimulEval-compatible bookkeeping around the agent.

SimulEval (the server/client that drives FairseqSimulSTAgent and scores the run) is third-party and absent from the
images; this module mirrors the parts of its contract the reference touches:
  * `units_to_segment` (agents/default_agent.py:248-301): merge committed subword units into words -- the server scores
    delays per WORD, not per subword
  * per-instance log + corpus scores in the schema of docs/mma.md:44-56
    ({"Quality": {"BLEU"}, "Latency": {"AL", "AL_CA", "AP", "AP_CA", "DAL", "DAL_CA"}})
  * computation-aware delays: source time at commit + wall-clock spent computing so far.
`run_instance` drives an agent with the same READ/WRITE protocol as agent.run_utterance and returns the instance log.
"""
import time
from typing import List, Optional, Sequence

from .latency import average_lagging, average_proportion, differentiable_average_lagging

BOW_PREFIX = "▁"
DEFAULT_EOS = "</s>"


class Dictionary:
    """The slice of fairseq's Dictionary the agent uses (eos(), string())."""

    def __init__(self, symbols: Sequence[str], eos_index: int = 2):
        self.symbols, self._eos = list(symbols), eos_index

    def eos(self):
        return self._eos

    def string(self, tokens, bpe_symbol: Optional[str] = None):
        s = " ".join(self.symbols[int(t)] for t in tokens if int(t) != self._eos)
        if bpe_symbol == "sentencepiece":
            s = s.replace(" ", "").replace(BOW_PREFIX, " ").strip()
        return s


class ListEntry:
    """simuleval.states.ListEntry as the reference uses it: a FIFO over .value (pop() removes the first element)."""

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


def units_to_segment(unit_queue: ListEntry, tgt_dict: Dictionary, n_target: int, max_len: float, pre_tokenizer=None):
    """agents/default_agent.py:248-301: returns DEFAULT_EOS, [hyp, DEFAULT_EOS], [word], [word, DEFAULT_EOS] or None."""
    if tgt_dict.eos() == unit_queue[0]:
        return DEFAULT_EOS
    segment: List[str] = []
    if None in unit_queue.value:
        unit_queue.value.remove(None)
    if (len(unit_queue) > 0 and tgt_dict.eos() == unit_queue[-1]) or n_target > max_len:
        hyp = tgt_dict.string(unit_queue.value, "sentencepiece")
        if pre_tokenizer is not None:
            hyp = pre_tokenizer.decode(hyp)
        return [hyp] + [DEFAULT_EOS]
    for index in list(unit_queue.value):
        token = tgt_dict.string([index])
        if token.startswith(BOW_PREFIX):
            if len(segment) == 0:
                segment += [token.replace(BOW_PREFIX, "")]
            else:
                for _ in range(len(segment)):
                    unit_queue.pop()
                out = ["".join(segment)]
                if tgt_dict.eos() == unit_queue[0]:
                    out += [DEFAULT_EOS]
                return out
        else:
            segment += [token.replace(BOW_PREFIX, "")]
    return None


def latency_scores(delays: Sequence[float], elapsed: Sequence[float], src_len_ms: float):
    out = {}
    for tag, d in (("", delays), ("_CA", elapsed)):
        out["AL" + tag] = average_lagging(d, src_len_ms)
        out["AP" + tag] = average_proportion(d, src_len_ms)
        out["DAL" + tag] = differentiable_average_lagging(d, src_len_ms)
    return out


def corpus_bleu(hyps: Sequence[str], refs: Sequence[str], max_n: int = 4) -> float:
    """Corpus BLEU-4 on whitespace tokens with the standard brevity penalty (sacrebleu, which the reference's
    SimulEval run uses, is absent: same formula, its 13a tokenizer not reproduced)."""
    import math
    from collections import Counter
    match, total = [0] * max_n, [0] * max_n
    hyp_len = ref_len = 0
    for h, r in zip(hyps, refs):
        ht, rt = h.split(), r.split()
        hyp_len, ref_len = hyp_len + len(ht), ref_len + len(rt)
        for n in range(1, max_n + 1):
            hc = Counter(tuple(ht[i:i + n]) for i in range(len(ht) - n + 1))
            rc = Counter(tuple(rt[i:i + n]) for i in range(len(rt) - n + 1))
            match[n - 1] += sum(min(c, rc[g]) for g, c in hc.items())
            total[n - 1] += max(len(ht) - n + 1, 0)
    if min(total) == 0 or min(match) == 0:
        return 0.0
    logp = sum(math.log(m / t) for m, t in zip(match, total)) / max_n
    bp = 1.0 if hyp_len > ref_len else math.exp(1.0 - ref_len / max(hyp_len, 1))
    return 100.0 * bp * math.exp(logp)


def run_instance(agent, fbank, tgt_dict: Dictionary, index: int = 0, reference: Optional[str] = None):
    """One utterance through `agent` (agent.FairseqSimulSTAgent) with word merging and SimulEval-style logging:
    delays / elapsed are per emitted WORD (the server's unit), stamped when units_to_segment releases it."""
    from .agent import READ_ACTION, FrameSource, States
    src = FrameSource(fbank)
    states = States(src)
    agent.initialize_states(states)
    queue = ListEntry()
    words, delays, elapsed = [], [], []
    t0 = time.perf_counter()
    done = False
    while not done:
        action = agent.policy(states)
        if action == READ_ACTION:
            if src.finished:
                raise RuntimeError("READ after source finished")
            src.read(agent.expected_frames)
            agent.update_states_read(states)
            continue
        tok = agent.predict(states)
        if tok is None:
            continue
        states.target.append(tok)
        agent.model.decoder.commit(states.dec_incremental_states["dec"])
        queue.append(tok)
        while len(queue) > 0:
            seg = units_to_segment(queue, tgt_dict, len(states.target), agent.max_len(src.pos))
            if seg is None:
                break
            now_ms = src.elapsed_ms()
            wall_ms = (time.perf_counter() - t0) * 1e3
            items = [seg] if isinstance(seg, str) else seg
            for w in items:
                if w == DEFAULT_EOS:
                    done = True
                else:
                    words.append(w)
                    delays.append(now_ms)
                    elapsed.append(now_ms + wall_ms)
            if done or isinstance(seg, str) or DEFAULT_EOS in items:
                done = True
                break
    prediction = " ".join(words)
    return {"index": index, "prediction": prediction, "delays": delays, "elapsed": elapsed,
            "prediction_length": len(words), "reference": reference, "source_length": src.total_ms(),
            "metric": {"latency": latency_scores(delays, elapsed, src.total_ms())}}


def corpus_scores(instances, references: Optional[Sequence[str]] = None):
    """docs/mma.md:44-56 schema: averages of the per-instance latency metrics (+ BLEU when references are given)."""
    keys = ("AL", "AL_CA", "AP", "AP_CA", "DAL", "DAL_CA")
    n = max(len(instances), 1)
    lat = {k: sum(i["metric"]["latency"][k] for i in instances) / n for k in keys}
    bleu = corpus_bleu([i["prediction"] for i in instances], references) if references is not None else None
    return {"Quality": {"BLEU": bleu}, "Latency": lat}
