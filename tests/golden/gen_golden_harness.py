#!/usr/bin/env python3
"""Record g15_units_to_segment.json by RUNNING THE REFERENCE's FairseqSimulSTAgent.units_to_segment
(agents/default_agent.py:248-301): subword queue -> words for the SimulEval server.

    python tests/golden/gen_golden_harness.py

The method is taken from the class imported by file path from /root/reference (never copied) and called with a bare
`self` carrying what it reads (dict["tgt"], pre_tokenizer, max_len).  fairseq and simuleval are absent from the
images; the stand-ins below restate the two external behaviours the method relies on (recalled, not verifiable here):
  * fairseq Dictionary.string(tokens, "sentencepiece"): pieces joined by " ", EOS dropped, then
    .replace(" ", "").replace("\\u2581", " ").strip()
  * simuleval ListEntry: a FIFO over `.value` -- pop() removes the FIRST element
"""
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden_fbank import load_reference_agent_module  # noqa: E402

BOW = "▁"


class Dictionary:
    def __init__(self, pieces):
        self.symbols = ["<s>", "<pad>", "</s>", "<unk>"] + list(pieces)

    def eos(self):
        return 2

    def string(self, tokens, bpe_symbol=None):
        toks = [self.symbols[int(t)] for t in tokens if int(t) != self.eos()]
        s = " ".join(toks)
        if bpe_symbol == "sentencepiece":
            s = s.replace(" ", "").replace(BOW, " ").strip()
        return s


class ListEntry:
    def __init__(self, value=None):
        self.value = list(value or [])

    def __len__(self):
        return len(self.value)

    def __getitem__(self, i):
        return self.value[i]

    def __iter__(self):
        return iter(list(self.value))

    def append(self, v):
        self.value.append(v)

    def pop(self, index=0):
        return self.value.pop(index)


PIECES = [BOW + "he", "llo", BOW + "wor", "ld", BOW + "a", BOW + "stream", "ing", BOW + "trans", "la", "tion", "s",
          BOW + "is", BOW + "here", "!", BOW + "x"]


def main():
    ref = load_reference_agent_module()
    fn = ref.FairseqSimulSTAgent.units_to_segment
    d = Dictionary(PIECES)
    rng = random.Random(999)
    cases = []
    for case in range(40):
        max_len = rng.choice([4, 8, 100])
        agent = types.SimpleNamespace(dict={"tgt": d}, pre_tokenizer=None, max_len=lambda src_len, m=max_len: m)
        n = rng.randint(1, 14)
        toks = [rng.randrange(4, len(d.symbols)) for _ in range(n)]
        if rng.random() < 0.7:
            toks.append(d.eos())
        if case == 0:
            toks = [d.eos()]
        if case == 1:
            toks = [None, 4, 5, 6, d.eos()]          # a force-finish None in the queue
        queue, target, events = ListEntry(), [], []
        for t in toks:
            queue.append(t)
            target.append(t)
            states = types.SimpleNamespace(units=types.SimpleNamespace(source=[0] * 50, target=target))
            out = fn(agent, queue, states)
            events.append({"pushed": t, "returned": out, "queue_after": list(queue.value)})
            if isinstance(out, list) and out and out[-1] == ref.DEFAULT_EOS or out == ref.DEFAULT_EOS:
                break
        cases.append({"max_len": max_len, "tokens": toks, "events": events})
    with open(os.path.join(HERE, "g15_units_to_segment.json"), "w") as f:
        json.dump({"pieces": PIECES, "eos": d.eos(), "default_eos": ref.DEFAULT_EOS, "cases": cases}, f, indent=0)
    print("g15_units_to_segment:", len(cases), "cases;", cases[3]["events"][:4])


if __name__ == "__main__":
    main()
