"""FairseqSimulSTAgent on MI355X: mirror of agents/default_agent.py (wait-k / MMA policies).

Same method names and control flow as the reference agent -- ``initialize_states``,
``update_states_read``, ``update_model_encoder``, ``policy``, ``predict`` -- with the model calls
going to the HIP encoder/decoder.  SimulEval (the process that drives the agent through
READ/WRITE actions) is third-party and absent from the images; ``FrameSource`` + ``run_utterance``
below stand in for its client/server loop at fbank-frame granularity (DESIGN.md "Synthetic
harness"): a READ releases the next ``expected_frames`` frames (fewer at the end, raising
``finish_read`` with the last ones) and every committed token is stamped with the source
milliseconds released so far.
"""
from typing import List, Optional

import torch

READ_ACTION, WRITE_ACTION = 0, 1
SHIFT_SIZE, WINDOW_SIZE = 10, 25


class FrameSource:
    def __init__(self, fbank: torch.Tensor):
        self.fbank = fbank
        self.pos = 0
        self.finished = fbank.size(0) == 0

    def read(self, n: int):
        self.pos = min(self.pos + n, self.fbank.size(0))
        self.finished = self.pos >= self.fbank.size(0)

    def elapsed_ms(self):
        return 0 if self.pos == 0 else self.pos * SHIFT_SIZE + (WINDOW_SIZE - SHIFT_SIZE)

    def total_ms(self):
        return self.fbank.size(0) * SHIFT_SIZE + (WINDOW_SIZE - SHIFT_SIZE)


class States:
    """The slice of SimulEval's SpeechStates the agent touches."""

    def __init__(self, source: FrameSource):
        self.source = source
        self.target: List[Optional[int]] = []
        self.enc_incremental_states = {}
        self.dec_incremental_states = {}
        self.last_update_source_len = 0

    def finish_read(self):
        return self.source.finished


class FairseqSimulSTAgent:
    def __init__(self, model, max_len_a: float = 1, max_len_b: int = 0, force_finish: bool = False):
        self.model = model
        enc = model.encoder
        # agents/default_agent.py:157-175
        self.pre_decision_ratio = model.decoder.pre_decision_ratio
        self.stride_ms = enc.conv_layer_stride() * SHIFT_SIZE
        self.right_context, self.segment_length = enc.right_context, enc.segment_length
        self.max_len = lambda x: min(max_len_a * x + max_len_b, model.max_decoder_positions())
        self.force_finish = force_finish
        self.eos = model.cfg.eos

    def initialize_states(self, states: States):
        states.enc_incremental_states = {}
        states.dec_incremental_states = {}

    # ---- agents/default_agent.py:303-342
    def update_model_encoder(self, states: States):
        src = states.source
        update_len = src.pos - states.last_update_source_len
        if update_len == 0 and states.finish_read():
            return
        finish = (update_len < self.expected_frames) or states.finish_read()
        frames = src.fbank[:src.pos].unsqueeze(0)
        out = self.model.encoder.infer(frames, torch.tensor([src.pos]), states.enc_incremental_states, finish=finish)
        new = out["encoder_out_btd"]
        dec = self.model.decoder
        if "dec" not in states.dec_incremental_states:
            # sized from what this source knows about itself; a longer stream makes the caches grow (DecoderState.grow)
            cap = int(self.max_len(src.fbank.size(0))) + 4
            s_cap = (src.fbank.size(0) // self.model.encoder.stride) + 2 * self.right_context + 8
            states.dec_incremental_states["dec"] = dec.new_state(1, cap=max(cap, 8), S_cap=max(s_cap, 8))
        st = states.dec_incremental_states["dec"]
        dec.append_encoder_out(st, new, torch.tensor([st.enc_rows + new.size(1)]))
        states.has_encoder_states = True
        states.last_update_source_len = src.pos

    def update_states_read(self, states: States):
        self.update_model_encoder(states)

    # ---- agents/default_agent.py:364-413
    def policy(self, states: States):
        if not getattr(states, "has_encoder_states", False):
            self.expected_frames = (self.segment_length + self.right_context) * self.stride_ms // SHIFT_SIZE
            if states.finish_read():
                self.update_states_read(states)
            return READ_ACTION
        st = states.dec_incremental_states["dec"]
        st.online = not states.finish_read()
        last = ([self.eos] + [t for t in states.target if t is not None])[-1]
        logits, action = self.model.decoder.step(st, torch.tensor([last], device=self.model.device),
                                                 stop_on_read=True)
        states.decoder_out = logits
        if action == 0:
            self.expected_frames = self.segment_length * self.stride_ms // SHIFT_SIZE
            return READ_ACTION
        return WRITE_ACTION

    # ---- agents/default_agent.py:415-436
    def predict(self, states: States):
        lprobs = self.model.get_normalized_probs([states.decoder_out], log_probs=True)
        index = int(lprobs.argmax(dim=-1)[0].item())
        if self.force_finish and index == self.eos and not states.finish_read():
            self.model.decoder.clear_cache(states.dec_incremental_states["dec"])
            return None
        return index

    # ---- the loop SimulEval runs around the agent
    def run_utterance(self, fbank: torch.Tensor):
        from .latency import average_lagging
        src = FrameSource(fbank)
        states = States(src)
        self.initialize_states(states)
        actions, delays = [], []
        while True:
            action = self.policy(states)
            if action == READ_ACTION:
                actions.append("R")
                if src.finished:
                    if not getattr(states, "has_encoder_states", False):
                        # the source ended before a single encoder state existed (empty / sub-frame input): the
                        # reference's policy keeps answering READ and SimulEval closes the sentence (default_agent.py:
                        # 370-376); here the hypothesis simply ends empty
                        actions.pop()
                        break
                    raise RuntimeError("READ after source finished")
                src.read(self.expected_frames)
                self.update_states_read(states)
                continue
            actions.append("W")
            tok = self.predict(states)
            if tok is None:
                continue
            states.target.append(tok)
            self.model.decoder.commit(states.dec_incremental_states["dec"])
            delays.append(src.elapsed_ms())
            # units_to_segment termination (agents/default_agent.py:268-271)
            if tok == self.eos or len(states.target) > self.max_len(src.pos):
                break
        st = states.dec_incremental_states.get("dec")
        return {"tokens": list(states.target), "delays_ms": delays, "actions": "".join(actions),
                "AL": average_lagging(delays, src.total_ms()), "n_enc": st.enc_rows if st is not None else 0}


def self_paced_records(hyp, delays, tok_chunk, n_prev, chunk_idx, cap, total_ms, extra_key, extra_of):
    """Host records of a self-paced run: the READ / WRITE string is rebuilt from the chunk index stamped on every token ("R" for
    every chunk the row took, then the tokens written at it); Average Lagging for all rows at once (latency.average_lagging_batch,
    bit-identical to the scalar function)."""
    import numpy as np
    from .latency import average_lagging_batch
    n_prev = np.minimum(n_prev.cpu().numpy(), cap)
    ci = chunk_idx.cpu().numpy()
    hyp_h, dl, tc = hyp.cpu().numpy(), delays.cpu().numpy(), tok_chunk.cpu().numpy()
    al = average_lagging_batch(dl, n_prev, total_ms)
    recs = []
    for b in range(len(n_prev)):
        n = int(n_prev[b])
        counts = np.bincount(tc[b, :n], minlength=int(ci[b]) + 1)
        recs.append({"tokens": hyp_h[b, :n].tolist(), "delays_ms": dl[b, :n].tolist(),
                     "actions": "".join("R" + "W" * int(c) for c in counts), "AL": al[b], extra_key: extra_of(b, int(ci[b]))})
    return recs


class BatchedStreamingAgent(FairseqSimulSTAgent):
    """B simultaneous streams through ONE encoder/decoder batch, each row taking its own READ / WRITE
    decisions.  The reference streams one utterance per process (models/s2t_emformer.py:200 asserts B == 1);
    this is the MI355X-shaped version of the same protocol: the sources advance in lockstep (every stream is
    offered the next chunk at the same time, as microphones do), between chunks the decoder repeats masked
    steps (simulst_mma_stream_steps) until every row is either waiting for source or finished.  Because a
    row's decisions depend only on its own state and on the source released so far, each row's tokens,
    delays and action string are those of ``FairseqSimulSTAgent.run_utterance`` on that utterance alone
    (tests/test_hip_streaming.py::test_batched_streaming_*).
    """

    def __init__(self, model, max_len_a: float = 1, max_len_b: int = 0, steps_per_call: int = 4, compact_rows: int = 0):
        """compact_rows > 0 (microphone form; round 6): every masked round runs over that many SLOTS instead of over all B rows --
        the rows that take part in the round are listed on the device first (simulst_stream_ctl.row_map), at most compact_rows of
        them, the others wait a round.  A round of a big group of live streams then costs what its ACTIVE rows cost (a parked row
        waits for its next chunk), so one group can hold thousands of streams; records are unchanged."""
        super().__init__(model, max_len_a, max_len_b, force_finish=False)
        self.steps_per_call = steps_per_call
        self.compact_rows = compact_rows

    def run_batch(self, fbank: torch.Tensor, self_paced: bool = False, encoder: str = "chunked", lengths=None):
        """fbank [B, T, 80] (equal lengths).  Returns one record per row, same keys as run_utterance.

        self_paced=False: the microphone form described above (one host round trip per chunk and per batch of masked steps).
        self_paced=True: the evaluation form (SimulEval feeding agents from files, agents/default_agent.py:303-342): every chunk
        goes through the streaming encoder first, then ONE device loop decodes, each row taking its next chunk by itself when its
        policy says READ (simulst_stream_ctl, self-paced rows) -- at most cap + n_chunks steps instead of the sum over chunks of the
        slowest row's steps.  Same records: a row's decisions depend only on its own state and the source released to it.
        encoder="offline" (self-paced only): the encoder states come from ONE offline forward over the whole source instead of
        the chunk-by-chunk ``infer`` calls -- the two agree to rounding (the reference's own check, agents/default_agent.py:438-476,
        atol = rtol = 1e-3; tests/test_hip_properties.py::test_full_size_streaming_equals_offline_fp32), the rows released per
        chunk are those of the streaming schedule (``stream_row_schedule``); decisions can differ from the chunked run only where
        that rounding flips one.  With it the sources may have different lengths (``lengths`` [B] frame counts, fbank padded): every
        row follows the chunk schedule of its own length."""
        if self_paced:
            return self._run_batch_self_paced(fbank, encoder, lengths)
        if lengths is not None and len({int(x) for x in lengths}) > 1:
            raise ValueError("sources of different lengths need self_paced=True, encoder='offline'")
        if encoder != "chunked":
            raise ValueError("the lockstep form streams through encoder.infer; encoder='offline' needs self_paced=True")
        return self._run_batch_lockstep(fbank)

    def _chunk_positions(self, T: int):
        """Source frames offered after each READ (agents/default_agent.py:367,407: first segment + right context, then segments)."""
        pos, out = 0, []
        expected = (self.segment_length + self.right_context) * self.stride_ms // SHIFT_SIZE
        while pos < T:
            pos = min(pos + expected, T)
            out.append(pos)
            expected = self.segment_length * self.stride_ms // SHIFT_SIZE
        return out

    def _run_batch_self_paced(self, fbank: torch.Tensor, encoder: str = "chunked", lengths=None):
        from . import _lib
        model, dec, enc = self.model, self.model.decoder, self.model.encoder
        cfg, dev = model.cfg, model.device
        B, T = fbank.size(0), fbank.size(1)
        fbank = fbank.to(dev)
        if encoder not in ("chunked", "offline"):
            raise ValueError(f"encoder={encoder!r}: 'chunked' or 'offline'")
        Ls = [T] * B if lengths is None else [int(x) for x in lengths]
        if len(Ls) != B or min(Ls) <= 0 or max(Ls) > T:
            raise ValueError("lengths: one positive frame count per row, at most fbank.size(1)")
        if len(set(Ls)) > 1 and encoder != "offline":
            raise ValueError("sources of different lengths need encoder='offline' (encoder.infer advances its rows in lockstep)")
        # ---- every row's schedule: frames offered after each READ, encoder rows released, source time, length cap
        #      (integer arithmetic per distinct length, cached on the agent; the [B, n_chunks] tables assembled with numpy)
        import numpy as np
        cache = self.__dict__.setdefault("_schedule_cache", {})
        per_T = {}
        for t in set(Ls):
            if t not in cache:
                positions = self._chunk_positions(t)
                cache[t] = (positions, enc.stream_row_schedule(positions),
                            [p * SHIFT_SIZE + (WINDOW_SIZE - SHIFT_SIZE) for p in positions], [int(self.max_len(p)) for p in positions])
            per_T[t] = cache[t]
        n_chunks = max(len(v[0]) for v in per_T.values())
        uniq, inv = np.unique(np.asarray(Ls), return_inverse=True)
        tab = np.zeros((3, len(uniq), n_chunks), dtype=np.int32)
        for u, t in enumerate(uniq.tolist()):
            for k in (1, 2, 3):
                v = per_T[t][k]
                tab[k - 1, u, :len(v)] = v
                tab[k - 1, u, len(v):] = v[-1]
        i32 = dict(device=dev, dtype=torch.int32)
        sched = torch.from_numpy(np.ascontiguousarray(tab[:, inv])).to(dev)                       # [3][B][n_chunks]
        row_chunks = torch.from_numpy(np.array([len(per_T[t][0]) for t in uniq.tolist()], dtype=np.int32)[inv]).to(dev)
        rows_total = [per_T[t][1][-1] for t in Ls]
        cap = int(self.max_len(max(Ls))) + 4
        st = dec.new_state(B, cap=cap, S_cap=max(max(rows_total), 1))
        st.lockstep = False
        if encoder == "offline":
            e = enc.forward(fbank, torch.tensor(Ls, device=dev))
            out = e["encoder_out_btd"]
            if e["encoder_lengths"].tolist() != rows_total:
                raise RuntimeError("offline encoder lengths differ from the rows the streaming schedule releases")
            dec.append_encoder_out(st, out, e["encoder_lengths"])
        else:
            # every chunk through the streaming encoder (the launches of the microphone form, minus its host round trips)
            enc_state, positions, rows = {}, per_T[T][0], []
            for i, pos in enumerate(positions):
                new = enc.infer(fbank[:, :pos], torch.full((B,), pos), enc_state, finish=i == len(positions) - 1)["encoder_out_btd"]
                dec.append_encoder_out(st, new, torch.full((B,), st.enc_rows + new.size(1)))
                rows.append(st.enc_rows)
            if rows != list(per_T[T][1]):
                raise RuntimeError(f"streaming encoder released {rows}, stream_row_schedule predicted {per_T[T][1]}")
        # wait-k: READ is a closed form of position and source length (position u needs u + k pooled keys), so every row starts at
        # the first chunk that allows position 0 and the commit kernel keeps taking chunks the same way (simulst_stream_ctl.ff_waitk)
        ff_k = cfg.waitk_lagging if cfg.attn_type == "waitk" else 0
        ratio_abs, pool_last = abs(dec.ratio_arg), dec.ratio_arg < 0

        def pooled_count(n):                                     # csrc/common.h pooled_count, incremental
            if pool_last and n < ratio_abs:
                return max(n - 1, 1)
            return min(-(-n // ratio_abs), max(n // ratio_abs, 1))
        c0_of = {}
        for t in uniq.tolist():
            rows_t, c = per_T[t][1], 0
            while ff_k > 0 and c + 1 < len(rows_t) and ff_k - 1 >= pooled_count(rows_t[c]):
                c += 1
            c0_of[t] = c
        c0 = np.array([c0_of[t] for t in uniq.tolist()], dtype=np.int32)[inv]
        chunk_idx = torch.from_numpy(c0.copy()).to(dev)
        st.enc_len = torch.from_numpy(np.ascontiguousarray(tab[0][inv, c0])).to(dev)
        st.enc_len_bh = st.enc_len.repeat_interleave(cfg.num_heads).contiguous()
        u8 = dict(device=dev, dtype=torch.uint8)
        active, read_flag, done = torch.ones(B, **u8), torch.zeros(B, **u8), torch.zeros(B, **u8)
        online = (chunk_idx + 1 < row_chunks).to(torch.uint8)
        hyp = torch.zeros(B, cap, device=dev, dtype=torch.int64)
        delays, tok_chunk = torch.zeros(B, cap, **i32), torch.zeros(B, cap, **i32)
        tokens = torch.full((B,), cfg.eos, device=dev, dtype=torch.int64)
        ctl = _lib.StreamCtl(active.data_ptr(), read_flag.data_ptr(), online.data_ptr(), done.data_ptr(), delays.data_ptr(),
                             hyp.data_ptr(), cap, 0, 0, n_chunks, sched[0].data_ptr(), sched[1].data_ptr(), sched[2].data_ptr(),
                             chunk_idx.data_ptr(), st.enc_len.data_ptr(), tok_chunk.data_ptr(), row_chunks.data_ptr(),
                             ff_k, dec.ratio_arg if ff_k else 0)
        # ---- one device loop; a row needs at most (tokens it may hold + 1) + (its chunks - 1) rounds
        bound = max(per_T[t][3][-1] + 1 + len(per_T[t][0]) for t in set(Ls))
        # no row can finish its cap in fewer rounds than it may hold tokens: that many rounds go out before the first look at the
        # masks (wait-k rows, whose READs cost no round, are then done unless they met EOS earlier -- which the look finds), short
        # calls after it
        first = max(per_T[t][3][-1] + 1 for t in set(Ls))
        n_run = 0
        while True:                                        # rows that meet EOS early end the loop before the bound
            n = max(1, min(64, first - n_run) if n_run < first else min(8, bound - n_run))
            dec.stream_steps(st, tokens, ctl, n)
            n_run += n
            if not bool(active.any().item()):
                break
            if n_run >= bound:
                raise RuntimeError("self-paced rows still active after cap + n_chunks rounds")
        return self_paced_records(hyp, delays, tok_chunk, st.n_prev, chunk_idx, cap,
                                  [t * SHIFT_SIZE + (WINDOW_SIZE - SHIFT_SIZE) for t in Ls], "n_enc",
                                  lambda b, c: per_T[Ls[b]][1][c])

    def _run_batch_lockstep(self, fbank: torch.Tensor):
        from . import _lib
        from .latency import average_lagging
        model, dec, enc = self.model, self.model.decoder, self.model.encoder
        cfg, dev = model.cfg, model.device
        B, T = fbank.size(0), fbank.size(1)
        fbank = fbank.to(dev)
        cap = int(self.max_len(T)) + 4
        # exactly the rows the schedule releases: up to 256 keys the policy / cross-attention kernel keeps its single-latency form
        s_cap = max(enc.stream_row_schedule(self._chunk_positions(T))[-1], 1)
        st = dec.new_state(B, cap=cap, S_cap=s_cap)
        st.lockstep = False
        u8 = dict(device=dev, dtype=torch.uint8)
        active, read_flag = torch.zeros(B, **u8), torch.zeros(B, **u8)
        online, done = torch.ones(B, **u8), torch.zeros(B, **u8)
        hyp = torch.zeros(B, cap, device=dev, dtype=torch.int64)
        delays = torch.zeros(B, cap, device=dev, dtype=torch.int32)
        tokens = torch.full((B,), cfg.eos, device=dev, dtype=torch.int64)
        # active-row compaction: slots per round (the kernels want more than 128 -- the layer chains' row class -- and at most B)
        n_slots = min(int(self.compact_rows), B) if self.compact_rows and B > 144 else 0
        if n_slots and n_slots <= 128:
            raise ValueError("compact_rows: more than 128 slots per round (or 0: off)")
        row_map = torch.empty(n_slots, device=dev, dtype=torch.int32) if n_slots else None
        enc_state = {}
        src = FrameSource(fbank[0])
        actions = [[] for _ in range(B)]
        n_written = [0] * B
        last_update = 0
        expected = (self.segment_length + self.right_context) * self.stride_ms // SHIFT_SIZE
        alive = list(range(B))
        while alive:
            # ---- READ phase: every unfinished stream takes the next chunk (agents/default_agent.py:303-342)
            for b in alive:
                actions[b].append("R")
            if src.finished:
                raise RuntimeError("READ after source finished")
            src.read(expected)
            update_len = src.pos - last_update
            finish = update_len < expected or src.finished
            out = enc.infer(fbank[:, :src.pos], torch.full((B,), src.pos), enc_state, finish=finish)
            new = out["encoder_out_btd"]
            dec.append_encoder_out(st, new, torch.full((B,), st.enc_rows + new.size(1)))
            last_update = src.pos
            expected = self.segment_length * self.stride_ms // SHIFT_SIZE
            # ---- WRITE phase: masked decoder steps until every row waits for source or is finished
            online.fill_(0 if src.finished else 1)
            active.copy_(1 - done)
            ctl = _lib.StreamCtl(active.data_ptr(), read_flag.data_ptr(), online.data_ptr(), done.data_ptr(),
                                 delays.data_ptr(), hyp.data_ptr(), cap, src.elapsed_ms(),
                                 int(self.max_len(src.pos)), 0, None, None, None, None, None, None, None, 0, 0,
                                 row_map=row_map.data_ptr() if row_map is not None else None, compact_rows=n_slots)
            while True:
                dec.stream_steps(st, tokens, ctl, self.steps_per_call)
                if not bool(active.any().item()):
                    break
            n_prev = st.n_prev.tolist()
            done_h = done.tolist()
            for b in alive:
                actions[b].extend("W" * (n_prev[b] - n_written[b]))
                n_written[b] = n_prev[b]
            alive = [b for b in alive if not done_h[b]]
        hyp_h, delays_h = hyp.tolist(), delays.tolist()
        recs = []
        for b in range(B):
            n = n_written[b]
            d = [int(x) for x in delays_h[b][:n]]
            recs.append({"tokens": hyp_h[b][:n], "delays_ms": d, "actions": "".join(actions[b]),
                         "AL": average_lagging(d, src.total_ms()), "n_enc": st.enc_rows})
        return recs


class ConcurrentStreamingEval:
    """Streaming evaluation of a test set: C self-paced batches in flight on C HIP streams, one host thread each -- the streaming
    counterpart of model.ConcurrentOffline (the reference evaluates one utterance per SimulEval process; a test set has thousands
    of independent ones, BASELINE.json configs[4]).  Every batch is ``BatchedStreamingAgent.run_batch(fbank, self_paced=True,
    encoder=..., lengths=...)`` (or the CIF agent's) on a replica that shares the device weights and owns its stream, handle and
    states, so a row's record does not depend on what rides beside it."""

    def __init__(self, model, weights, concurrency: int = 3, agent_factory=None, model_factory=None):
        """agent_factory(model_replica) -> an agent with run_batch; model_factory(ops) -> a replica of another model class (the CIF
        model); defaults: BatchedStreamingAgent over SimulSTModel replicas that share ``model``'s device weights."""
        from . import _lib
        from .model import SimulSTModel
        from .ops import Ops
        self.agents, self.streams = [], []
        self.device = model.device
        for _ in range(max(1, concurrency)):
            st = torch.cuda.Stream(device=model.device)
            with torch.cuda.stream(st):
                ops = Ops(_lib.Handle(st.cuda_stream))
                m = model_factory(ops) if model_factory is not None else \
                    SimulSTModel(model.cfg, weights, device=model.device, dtype=model.dtype, ops=ops, share_with=model)
                self.agents.append(agent_factory(m) if agent_factory is not None else BatchedStreamingAgent(m))
            self.streams.append(st)

    def run(self, batches, encoder: str = "offline", self_paced: bool = True):
        """batches: (fbank [B, T, 80], lengths or None) pairs, already on the device.  Returns one list of records per batch.
        The streams take the batches from a queue in the given order (a run ends with a read-back, so a free host thread means a
        free stream): put the expensive ones first.
        self_paced=False: every batch in the MICROPHONE form (sources advance in lockstep, one host round trip per chunk and per group
        of masked steps; equal lengths, the streaming encoder): several groups of live streams side by side -- a group's masked steps
        are latency-bound whatever its row count, so groups on different HIP streams overlap."""
        import threading
        batches = list(batches)
        out, errs = [None] * len(batches), []
        queue, qlock = list(range(len(batches))), threading.Lock()
        cur = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(cur)
        dev_index = self.device.index

        def worker(c):
            try:
                if dev_index is not None:
                    torch.cuda.set_device(dev_index)
                with torch.no_grad(), torch.cuda.stream(self.streams[c]):
                    while True:
                        with qlock:
                            if not queue:
                                break
                            i = queue.pop(0)
                        fb, lengths = batches[i]
                        out[i] = (self.agents[c].run_batch(fb, self_paced=True, encoder=encoder, lengths=lengths) if self_paced
                                  else self.agents[c].run_batch(fb))
            except Exception as e:          # surfaced to the caller below
                errs.append(e)

        threads = [threading.Thread(target=worker, args=(c,)) for c in range(len(self.agents))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for st in self.streams:             # host-side join (model.ConcurrentOffline.run explains why not wait_stream)
            st.synchronize()
        if errs:
            raise errs[0]
        return out
