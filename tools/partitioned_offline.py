"""Experiment (measured NEGATIVE, profiles/r05_cu_partition_sweep.txt): the encoder of the later launch sequences beside the decode loops of
the earlier ones on disjoint compute units.  Kept outside the product package; `bench.py --partition CUS` drives it."""
import os
import sys
from typing import Optional

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from simulst_amd.model import SimulSTModel  # noqa: E402
from simulst_amd.ops import Ops  # noqa: E402


class PartitionedOffline:
    """The offline evaluation loop (eval/generate.py:187-209 over independent batches) with the ENCODER of the later launch sequences
    running beside the DECODE loops of the earlier ones on disjoint compute units.

    Why masks: a decode loop is a chain of ~3 400 small dependent launches; beside a chip-filling matrix kernel of another stream each of
    them waits for a free slot (3.1 -> 12.4 us per 28-workgroup kernel, stream priorities change nothing: tools/microbench_cumask.hip),
    which is why chaining the encoder passes of plain streams lost (ConcurrentOffline(stagger_encoders=True)).  With the encoder
    confined to `32 - decode_cus_per_xcd` compute units of every XCD and the decode streams to the other `decode_cus_per_xcd` the chain
    runs at its own pace (3.4 us) and the encoder at its share of the chip.  Schedule for k launch sequences, k_a = `phase_a` of them
    in the first phase:
        encoder(sequences 0 .. k_a-1)   whole chip (nothing else is running yet)
        encoder(sequences k_a .. k-1)   encoder units    ||   decode(0 .. k_a-1)  decode units
        decode(k_a .. k-1)              whole chip (plain streams: the encoder units are idle by then)
    Every sequence runs exactly the kernels of SimulSTModel.generate_offline on its own rows (the encoder of several sequences is ONE
    pass over their stacked rows, as in ConcurrentOffline's joint pass: rows are independent), so results are identical."""

    def __init__(self, model: SimulSTModel, weights, concurrency: int = 4, decode_cus_per_xcd: int = 16, phase_a: Optional[int] = None):
        from simulst_amd import _lib
        self.model, self.phase_a = model, phase_a
        dev = model.device
        d = int(decode_cus_per_xcd)
        if not 1 <= d <= 31:
            raise ValueError("decode_cus_per_xcd: 1 .. 31 of an XCD's 32 compute units")
        self.decode_cus_per_xcd = d
        mask_d, mask_e = _lib.cu_mask_words(d), _lib.cu_mask_words(32 - d, take_high=True)
        self.s_enc_full = torch.cuda.Stream(device=dev)
        self.s_enc_part = _lib.create_stream(dev, cu_mask=mask_e)
        with torch.cuda.stream(self.s_enc_full):
            self.enc_ops = Ops(_lib.Handle(self.s_enc_full.cuda_stream))
            self.enc_model = SimulSTModel(model.cfg, weights, device=dev, dtype=model.dtype, ops=self.enc_ops, share_with=model)
        # decode replicas: one per launch sequence; their streams are made on first use (a masked one for a sequence of the first
        # phase, a plain one for the others) so that no more HIP streams exist than run -- the device keeps a handful of hardware queues
        # and streams that share one serialise
        self.models, self._streams = [], {}
        self._mask_d, self._weights = mask_d, weights
        for _ in range(concurrency):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                ops = Ops(_lib.Handle(st.cuda_stream))
                self.models.append(SimulSTModel(model.cfg, weights, device=dev, dtype=model.dtype, ops=ops, share_with=model))
            self._streams[(len(self.models) - 1, False)] = st

    # NOTE: the phase-A and phase-B encoder passes share ONE enc_model and its cached layer workspace and flip enc_ops.h between
    # the whole-chip stream and the CU-masked one; only phase B's wait on phase A's event keeps the two passes apart -- do not drop it.
    # Tensors allocated under an external stream are kept alive by the callers' local lists (no record_stream), and the masked HIP
    # streams are destroyed by close().  An experiment, measured negative (profiles/r05_cu_partition_sweep.txt); not part of the package.
    def close(self):
        from simulst_amd import _lib
        for (i, masked), st in list(self._streams.items()):
            if masked and hasattr(_lib, "destroy_stream"):
                _lib.destroy_stream(st)
                del self._streams[(i, masked)]

    def _stream(self, i, masked):
        from simulst_amd import _lib
        key = (i, bool(masked))
        if key not in self._streams:
            self._streams[key] = _lib.create_stream(self.model.device, cu_mask=self._mask_d) if masked else torch.cuda.Stream(device=self.model.device)
        return self._streams[key]

    def _encode(self, batches, stream, after=None):
        """ONE encoder pass over the stacked rows of `batches` on `stream` -> per batch (encoder rows, lengths), the event, the dict"""
        toks = [b[0] for b in batches]
        esz = toks[0].element_size()
        adjacent = all(t.is_contiguous() and t.shape[1:] == toks[0].shape[1:] for t in toks) and all(
            toks[i].data_ptr() + toks[i].numel() * esz == toks[i + 1].data_ptr() for i in range(len(toks) - 1))
        total = sum(t.size(0) for t in toks)
        self.enc_ops.h.set_stream(stream.cuda_stream)
        with torch.no_grad(), torch.cuda.stream(stream):           # the gathers on the encoder's stream (model.ConcurrentOffline._joint_encoder)
            if after is not None:
                stream.wait_event(after)
            if adjacent and toks[0].untyped_storage().data_ptr() == toks[-1].untyped_storage().data_ptr():
                allt = toks[0].as_strided((total,) + tuple(toks[0].shape[1:]), toks[0].stride())
            else:
                allt = torch.cat(toks, 0)
            lens = torch.cat([b[1].to(allt.device) for b in batches], 0)
            enc = self.enc_model.encoder.forward(allt, lens)
            ev = torch.cuda.Event()
            ev.record(stream)
        parts, r0 = [], 0
        for t in toks:
            parts.append((enc["encoder_out_btd"][r0:r0 + t.size(0)], enc["encoder_lengths"][r0:r0 + t.size(0)]))
            r0 += t.size(0)
        return parts, ev, enc

    def run(self, batches, n_steps: int, mask_eos: bool = False):
        import threading
        batches = list(batches)
        k = len(batches)
        if k > len(self.models):
            raise ValueError(f"{k} launch sequences for {len(self.models)} decode replicas")
        if len({tuple(b[0].shape[1:]) for b in batches}) != 1:
            raise ValueError("PartitionedOffline stacks the sequences' rows for the encoder: equal frame counts needed")
        k_a = min(k, self.phase_a if self.phase_a else (k + 1) // 2)
        cur = torch.cuda.current_stream()
        used = [self.s_enc_full, self.s_enc_part] + [self._stream(i, i < k_a) for i in range(k)]
        for st in used:
            st.wait_stream(cur)
        keep = []
        parts_a, ev_a, enc_a = self._encode(batches[:k_a], self.s_enc_full)
        keep.append(enc_a)
        parts, evs = list(parts_a), [ev_a] * k_a
        if k > k_a:
            parts_b, ev_b, enc_b = self._encode(batches[k_a:], self.s_enc_part, after=ev_a)
            keep.append(enc_b)
            parts += parts_b
            evs += [ev_b] * (k - k_a)
        out, errs = [None] * k, []
        dev_index = self.model.device.index

        def worker(i):
            try:
                if dev_index is not None:
                    torch.cuda.set_device(dev_index)
                st = self._stream(i, i < k_a)
                m = self.models[i]
                m.ops.h.set_stream(st.cuda_stream)
                with torch.no_grad(), torch.cuda.stream(st):
                    st.wait_event(evs[i])
                    e_out, e_len = parts[i]
                    out[i] = m.decoder.greedy_offline(e_out, e_len, n_steps, mask_eos)[0].clone()
            except Exception as e:          # surfaced to the caller below
                errs.append(e)

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(k)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for st in used:      # host-side join (see ConcurrentOffline.run)
            st.synchronize()
        del keep
        if errs:
            raise errs[0]
        return out
