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

    def pad(self):
        return 1

    def __len__(self):
        return len(self.symbols)

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


class WordMerger:
    """Subword units -> words for the SimulEval server, which scores delays per WORD (the contract of
    agents/default_agent.py:248-301).  Table-driven: per vocabulary id, whether the piece opens a word (starts with
    the sentencepiece marker) and its text without the marker, both built once; a call is then one scan for the first
    word-opening piece behind the head of the queue and one slice.

    Outcomes, in this order: the queue starts with EOS -> DEFAULT_EOS; (after dropping one force-finish ``None``) the
    queue ends with EOS, or the hypothesis has outgrown max_len -> [detokenised queue, DEFAULT_EOS] with the queue left
    as it is; a word-opening piece at position p >= 1 -> the p pieces before it leave the queue as one word (followed
    by DEFAULT_EOS when EOS is what remains in front); otherwise None: the word is still open."""

    def __init__(self, tgt_dict):
        self.dict = tgt_dict
        self.eos = tgt_dict.eos()
        n = len(tgt_dict) if hasattr(tgt_dict, "__len__") else len(tgt_dict.symbols)
        pieces = [tgt_dict.string([i]) for i in range(n)]
        self.opens_word = [p.startswith(BOW_PREFIX) for p in pieces]
        self.text = [p.replace(BOW_PREFIX, "") for p in pieces]

    def __call__(self, unit_queue, n_target: int, max_len: float, pre_tokenizer=None):
        units = unit_queue.value
        if units[0] == self.eos:
            return DEFAULT_EOS
        if None in units:
            units.remove(None)
        if (units and units[-1] == self.eos) or n_target > max_len:
            hyp = self.dict.string(units, "sentencepiece")
            return [pre_tokenizer.decode(hyp) if pre_tokenizer is not None else hyp, DEFAULT_EOS]
        cut = next((p for p in range(1, len(units)) if self.opens_word[units[p]]), None)
        if cut is None:
            return None
        word = "".join(self.text[t] for t in units[:cut])
        del units[:cut]
        return [word, DEFAULT_EOS] if units[0] == self.eos else [word]


_mergers = {}


def units_to_segment(unit_queue: ListEntry, tgt_dict, n_target: int, max_len: float, pre_tokenizer=None):
    """Function form of WordMerger (one merger per dictionary object, built on first use)."""
    m = _mergers.get(id(tgt_dict))
    if m is None or m.dict is not tgt_dict:
        m = _mergers[id(tgt_dict)] = WordMerger(tgt_dict)
    return m(unit_queue, n_target, max_len, pre_tokenizer)


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
