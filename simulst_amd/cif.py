"""CIF adaptive-policy model on MI355X: mirror of models/cif_transformer.py (CIFLayer, CIFEncoder,
FakeCrossAttn/CIFDecoderLayer/CIFDecoder) and agents/cif_agent.py.

The integrate-and-fire scan itself (``torch_cif.cif_function`` in the reference -- an un-vendored
submodule) is the HIP kernel ``simulst_cif_integrate``: wavefront prefix-sum of alpha, fire index
floor(csum / beta), segmented weighted sum in one pass over the source.
"""
import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from .agent import READ_ACTION, WRITE_ACTION, FrameSource, States
from .config import ModelConfig
from .decoder import DecoderWeights
from .encoder import S2TEmformerEncoder
from .ops import Ops, EPI_BIAS, EPI_BIAS_F32OUT, EPI_BIAS_GELU, EPI_BIAS_RES
from .model import FairseqModelSurface
from .registry import register_model, register_model_architecture


class CIFLayer:
    """models/cif_transformer.py:111-261 (alpha_proj = CausalConvTBC -> LN -> GELU -> Linear -> sigmoid)."""

    def __init__(self, cfg: ModelConfig, w: Dict[str, torch.Tensor], ops: Ops, device, dtype,
                 prefix="encoder.cif_layer"):
        self.cfg, self.ops, self.device, self.dtype = cfg, ops, device, dtype
        self.beta = cfg.cif_beta
        self.tail_thres = cfg.cif_beta / 2
        cw = w[prefix + ".alpha_proj.0.weight"]                 # ConvTBC [k, C_in, C_out]
        self.k = cw.shape[0]
        # GEMM over overlapping channel-last rows: W[o][tau*C + c] = cw[tau][c][o]
        # Round 6: the weight head runs in fp32 whatever the model's dtype -- fp32 copies of its two weight
        # matrices, the encoder rows promoted, the conv output kept in fp32.  The weights alpha are SUMMED over the source (the
        # integrate-and-fire scan), so an error that does not change sign from frame to frame accumulates linearly: the bf16 copies
        # of the head's weights alone put +0.042 on the accumulated weight of a 250-frame source, the same sign for every utterance
        # (tools/cif_alpha_budget.py, profiles/r06_cif_alpha_budget.json: 38 % of the path's mean signed error).  The head is
        # 0.1 GFLOP per utterance; the encoder (13.5 GFLOP) stays in the model's dtype.
        self.head_dtype = torch.float32 if self.head_fp32 else dtype
        self.conv_w = cw.permute(2, 0, 1).reshape(cw.shape[2], -1).contiguous().to(device=device, dtype=self.head_dtype)
        self.conv_b = w[prefix + ".alpha_proj.0.bias"].float().to(device)
        self.ln_g = w[prefix + ".alpha_proj.1.weight"].float().to(device)
        self.ln_b = w[prefix + ".alpha_proj.1.bias"].float().to(device)
        self.out_w = w[prefix + ".alpha_proj.4.weight"].reshape(-1).contiguous().to(device=device, dtype=self.head_dtype)
        self.out_b = float(w[prefix + ".alpha_proj.4.bias"][0])

    head_fp32 = os.environ.get("SIMULST_CIF_HEAD_FP32", "1") != "0"       # (0: the head in the model's dtype, rounds 1-5; A/B runs)

    def _alpha(self, x_btd: torch.Tensor, hist: Optional[torch.Tensor]):
        """sigmoid(alpha_proj(x)). x [B,T,D]; hist [B,k-1,D] = frames before x (None: zero left pad)."""
        ops = self.ops
        B, T, D = x_btd.shape
        k = self.k
        if hist is not None:
            xc = torch.cat([hist, x_btd], dim=1).to(self.head_dtype).contiguous()
            lead = 0
        else:
            xc = x_btd.to(self.head_dtype).contiguous()
            lead = (k - 1) * D
        Tc = xc.size(1)
        h = torch.empty(B, T, D, device=x_btd.device, dtype=self.head_dtype)
        ops.linear_raw(xc, self.conv_w, self.conv_b, h, M_batches=B, rows_per_batch=T, N=D, K=k * D, a_bs=Tc * D,
                       a_rs=D, a_lead=lead, c_bs=T * D, c_rs=D, epilogue=EPI_BIAS)
        return ops.cif_alpha_head(h, self.ln_g, self.ln_b, self.out_w, self.out_b)

    def forward(self, x_btd: torch.Tensor, lengths: torch.Tensor):
        """CIFLayer.forward (:141-186), inference (no target_lengths). x [B,T,D], lengths [B]."""
        alpha = self._alpha(x_btd, None)
        len_i32 = lengths.to(device=self.device, dtype=torch.int32)
        out, n, delays, tail_w, asum = self.ops.cif_integrate(x_btd.contiguous(), alpha, beta=self.beta,
                                                               tail_thres=self.tail_thres, src_len=len_i32)
        return {"cif_out": [out.transpose(0, 1)], "cif_lengths": [n.to(torch.int64)], "alpha": [alpha],
                "delays": [delays], "alpha_sum": [asum], "tail_weights": [tail_w], "cif_out_btd": out}

    def new_state(self, B: int, D: int):
        return {"conv_hist": torch.zeros(B, self.k - 1, D, device=self.device, dtype=self.dtype),
                "prev_feat": None, "prev_weight": None}

    def infer(self, x_btd: torch.Tensor, st: dict, finish: bool = False):
        """CIFLayer.infer (:188-261): B == 1; the un-fired tail (weight, feature / beta) is carried as a
        pseudo source frame in front of the next chunk."""
        B, T, D = x_btd.shape
        if B > 1:
            raise NotImplementedError("batched infer not supported for now.")
        if T > 0:
            alpha = self._alpha(x_btd, st["conv_hist"])
            st["conv_hist"] = torch.cat([st["conv_hist"], x_btd], dim=1)[:, -(self.k - 1):].contiguous()
        else:
            alpha = torch.zeros(B, 0, device=self.device, dtype=torch.float32)
        x = x_btd
        if st["prev_weight"] is not None:
            alpha = torch.cat([st["prev_weight"], alpha], dim=1).contiguous()
            x = torch.cat([st["prev_feat"], x], dim=1).contiguous()
        out, n, delays, tail_w, asum = self.ops.cif_integrate(x.contiguous(), alpha, beta=self.beta,
                                                               tail_thres=self.tail_thres if finish else 0.0)
        n_host = int(n[0].item())
        if not finish:
            st["prev_feat"] = (out[:, n_host - 1:n_host].float() / self.beta).to(self.dtype)
            st["prev_weight"] = tail_w.view(B, 1)
            n_host -= 1
        else:
            st["prev_feat"] = st["prev_weight"] = None
        feats = out[:, :n_host]
        return {"cif_out": [feats.transpose(0, 1)], "cif_lengths": [torch.tensor([n_host])], "alpha": [alpha],
                "delays": [delays], "alpha_sum": [asum], "tail_weights": [tail_w], "cif_out_btd": feats}


    # ---- batched streaming (no counterpart in the reference: CIFLayer.infer raises for B > 1, :199-200) -------------------
    def new_batched_state(self, B: int, D: int, n_cap: int):
        """per-row carry and accumulated integrated vectors of B lockstep streams, all on the device"""
        dev = self.device
        return {"conv_hist": torch.zeros(B, self.k - 1, D, device=dev, dtype=self.dtype), "first": True,
                "prev_feat": torch.zeros(B, 1, D, device=dev, dtype=self.dtype),
                "prev_weight": torch.zeros(B, 1, device=dev, dtype=torch.float32),
                "cif": torch.zeros(B, n_cap, D, device=dev, dtype=self.dtype), "n_cap": n_cap,
                "cif_len": torch.zeros(B, device=dev, dtype=torch.int32)}

    def infer_batched(self, x_btd: torch.Tensor, st: dict, finish: bool = False, trace: Optional[list] = None):
        """CIFLayer.infer for B rows that receive the same number of new encoder frames (lockstep sources): per row exactly
        the B == 1 arithmetic -- weights of the new frames, the carried (weight, feature / beta) pseudo-frame in front
        (:217-226), one integrate-and-fire scan, the un-fired tail withheld and carried unless ``finish`` (:235-255).  The new
        vectors are appended to st["cif"] at st["cif_len"] on the device (simulst_cif_stream_append); nothing is read back."""
        B, T, D = x_btd.shape
        if T > 0:
            alpha = self._alpha(x_btd, st["conv_hist"])
            st["conv_hist"] = torch.cat([st["conv_hist"], x_btd], dim=1)[:, -(self.k - 1):].contiguous()
        else:
            alpha = torch.zeros(B, 0, device=self.device, dtype=torch.float32)
        x = x_btd
        if not st["first"]:
            alpha = torch.cat([st["prev_weight"], alpha], dim=1).contiguous()
            x = torch.cat([st["prev_feat"], x], dim=1).contiguous()
        st["first"] = False
        if x.size(1) == 0:
            return
        out, n, _, tail_w, asum = self.ops.cif_integrate(x.contiguous(), alpha, beta=self.beta,
                                                         tail_thres=self.tail_thres if finish else 0.0)
        if trace is not None:         # parity audit (tools/teacher_forced_audit.py): what this call's scan accumulated and released
            trace.append({"alpha_sum": asum.clone(), "n": n.clone(), "tail": tail_w.clone(), "finish": finish})
        self.ops.cif_stream_append(out, n, tail_w, st["cif"], st["cif_len"], st["prev_feat"], st["prev_weight"],
                                   beta=self.beta, finish=finish)


class CIFEncoder(S2TEmformerEncoder):
    """models/cif_transformer.py:264-321."""

    def __init__(self, cfg, weights, device="cuda", dtype=torch.float32, ops=None):
        super().__init__(cfg, weights, device, dtype, ops)
        self.cif_layer = CIFLayer(cfg, weights, self.ops, self.device, dtype)

    def forward(self, src_tokens, src_lengths):
        enc = super().forward(src_tokens, src_lengths)
        enc.update(self.cif_layer.forward(enc["encoder_out_btd"].contiguous(), enc["encoder_lengths"]))
        return enc

    def infer(self, src_tokens, src_lengths, incremental_state, finish=False):
        enc = super().infer(src_tokens, src_lengths, incremental_state, finish=finish)
        key = "simulst_amd.cif_state"
        if key not in incremental_state:
            incremental_state[key] = self.cif_layer.new_state(src_tokens.size(0), self.cfg.embed_dim)
        enc.update(self.cif_layer.infer(enc["encoder_out_btd"].contiguous(), incremental_state[key], finish))
        return enc


class CIFDecoder:
    """models/cif_transformer.py:540-724: position-synchronous decoding; the 'cross attention' is
    FakeCrossAttn out_proj(gelu(q_proj(x) + k_proj(cif_t))) (:340-362)."""

    def __init__(self, cfg: ModelConfig, weights, device="cuda", dtype=torch.float32, ops=None, prefix="decoder"):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.ops = ops or Ops()
        self.w = DecoderWeights(weights, cfg, self.device, dtype, prefix)
        self.embed_scale = 1.0 if cfg.no_scale_embedding else math.sqrt(cfg.embed_dim)
        self.pre_decision_ratio = 1
        # what the agent reads off the decoder (agents/cif_agent.py:150-155,372-405,428-433)
        from types import SimpleNamespace
        from .decoder import _Dictionary
        self.dictionary = _Dictionary(cfg)
        self.layers = [SimpleNamespace(index=l, encoder_attn=SimpleNamespace()) for l in range(cfg.decoder_layers)]

        D, F, V = cfg.embed_dim, cfg.ffn_dim, cfg.vocab
        self.fragment_major = D % 64 == 0 and F % 64 == 0 and V % 16 == 0 and cfg.head_dim % 16 == 0
        import os
        self.layer_chains = os.environ.get("SIMULST_LAYER_CHAINS", "1") == "1"
        self._pack_all()

    def _pack_all(self):
        """decode-loop weight matrices in MFMA-fragment order (simulst_linear_desc.w_fragment_major), made once"""
        w = self.w
        w.packed, w.out_proj_packed, w.kc_packed = None, None, {}
        if self.fragment_major:
            names = ["wqkv", "wo", "c_wq", "c_wo", "fc1", "fc2"]
            w.packed = [{n: self.ops.pack_fragment_major(L[n]) for n in names} for L in w.layers]
            w.out_proj_packed = self.ops.pack_fragment_major(w.out_proj)
        if self.dtype == torch.bfloat16 and self.cfg.embed_dim <= 256 and self.cfg.embed_dim % 64 == 0:
            for L in w.layers:                      # the row-panel kernel's form of k_proj (thousands of integrated vectors)
                w.kc_packed[L["c_wk"].data_ptr()] = self.ops.pack_fragment_major(L["c_wk"])
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def clear_cache(self, incremental_state, end_id=None):
        """a discarded prediction needs no rollback: forward() takes the written-token count from prev_output_tokens"""
        return None

    # ------------------------------------------------------------------ device-resident loop (simulst_cif_decode)
    def new_device_state(self, B: int, cap: int, n_cap: int):
        cfg, dev, dt_ = self.cfg, self.device, self.dtype
        H, d, D, Ld = cfg.num_heads, cfg.head_dim, cfg.embed_dim, cfg.decoder_layers
        from .decoder import ensure_positions
        ensure_positions(self.w, cap + cfg.padding_idx + 2)
        st = {"B": B, "cap": cap, "n_cap": n_cap,
              "k": [torch.zeros(B, H, cap, d, device=dev, dtype=dt_) for _ in range(Ld)],
              "v": [torch.zeros(B, H, cap, d, device=dev, dtype=dt_) for _ in range(Ld)],
              "Kc": [torch.zeros(B, n_cap, D, device=dev, dtype=dt_) for _ in range(Ld)],
              "n_prev": torch.zeros(B, device=dev, dtype=torch.int32), "n_prev_host": 0, "lockstep": True,
              "cif": None, "cif_len": None, "tok": torch.empty(B, device=dev, dtype=torch.int64),
              "ws": {"x": torch.empty(B, D, device=dev, dtype=dt_), "qkv": torch.empty(B, 3 * D, device=dev, dtype=dt_),
                     "ctx": torch.empty(B, D, device=dev, dtype=dt_), "q": torch.empty(B, D, device=dev, dtype=dt_),
                     "hidden": torch.empty(B, cfg.ffn_dim, device=dev, dtype=dt_),
                     "logits": torch.empty(B, cfg.vocab, device=dev, dtype=torch.float32),
                     "kk": torch.empty(Ld, B, D, device=dev, dtype=dt_), "cif_t": torch.empty(B, D, device=dev, dtype=dt_),
                     "eos_bias": torch.zeros(B, device=dev, dtype=torch.float32),
                     "x_mid": torch.empty(B, D, device=dev, dtype=dt_)}}
        if self.layer_chains and dt_ == torch.bfloat16 and D == 256 and cfg.ffn_dim % 256 == 0 and self.fragment_major:
            st["ws"]["ffn_partial"] = torch.empty(cfg.ffn_dim // 256, B, D, device=dev, dtype=torch.float32)
        arr = (_lib.CifDecLayer * Ld)()
        for l, L in enumerate(self.w.layers):
            a, P = arr[l], (self.w.packed[l] if self.fragment_major else L)
            for n in ("bqkv", "bo", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "ln3_g", "ln3_b", "c_bo", "b1", "b2"):
                setattr(a, n, L[n].data_ptr())
            for n in ("wqkv", "wo", "c_wq", "c_wo", "fc1", "fc2"):
                setattr(a, n, P[n].data_ptr())
            a.k_cache, a.v_cache, a.Kc = st["k"][l].data_ptr(), st["v"][l].data_ptr(), st["Kc"][l].data_ptr()
        st["layer_structs"] = arr
        return st

    def project_cif(self, st, cif: torch.Tensor, lo: int, hi: int):
        """Kc_l[:, lo:hi] = k_proj_l(cif[:, lo:hi]) + bias for every layer: FakeCrossAttn's key projection (:358), done once per
        integrated vector instead of once per target position.  cif [B, n_cap, D] is kept as the state's vector buffer."""
        ops, D = self.ops, self.cfg.embed_dim
        st["cif"] = cif
        n = hi - lo
        if n <= 0:
            return
        B, n_cap = cif.size(0), cif.size(1)
        assert n_cap == st["n_cap"] and cif.is_contiguous()
        for l, L in enumerate(self.w.layers):
            Wt, fm = L["c_wk"], False
            if B * n >= 4096 and Wt.data_ptr() in self.w.kc_packed:
                Wt, fm = self.w.kc_packed[Wt.data_ptr()], True
            ops.linear_raw(cif[:, lo:], Wt, L["c_bk"], st["Kc"][l][:, lo:], M_batches=B, rows_per_batch=n, N=D, K=D,
                           a_bs=n_cap * D, a_rs=D, c_bs=n_cap * D, c_rs=D, epilogue=EPI_BIAS, w_fragment_major=fm)

    def _desc(self, st, np_uniform: int, overshoot_weight: float):
        cfg, ws = self.cfg, st["ws"]
        from .decoder import ensure_positions
        ensure_positions(self.w, st["cap"] + cfg.padding_idx + 2)
        out_proj = self.w.out_proj_packed if self.fragment_major else self.w.out_proj
        chains = "ffn_partial" in ws
        return _lib.CifDecoderDesc(st["B"], cfg.embed_dim, cfg.num_heads, cfg.ffn_dim, cfg.vocab, cfg.decoder_layers, st["cap"],
                                   st["n_cap"], _lib.F32 if self.dtype == torch.float32 else _lib.BF16, cfg.padding_idx, cfg.eos,
                                   int(cfg.cif_highway), np_uniform, self.embed_scale, float(overshoot_weight),
                                   self.w.E.data_ptr(), out_proj.data_ptr(), self.w.pos.data_ptr(), self.w.ln_g.data_ptr(),
                                   self.w.ln_b.data_ptr(), st["cif_len"].data_ptr(), st["cif"].data_ptr(), st["n_prev"].data_ptr(),
                                   ws["x"].data_ptr(), ws["qkv"].data_ptr(), ws["ctx"].data_ptr(), ws["q"].data_ptr(),
                                   ws["hidden"].data_ptr(), ws["logits"].data_ptr(), ws["kk"].data_ptr(), ws["cif_t"].data_ptr(),
                                   ws["eos_bias"].data_ptr(), ws["x_mid"].data_ptr() if chains else None,
                                   ws["ffn_partial"].data_ptr() if chains else None, int(self.fragment_major))

    def decode_steps(self, st, n_steps: int, mask_eos: bool, overshoot_weight: float = 1.0):
        """n_steps target positions of every row on the device (simulst_cif_decode); st["tok"] holds the newest token of
        [eos] + hyp and is updated in place.  Returns tokens [n_steps, B]."""
        import ctypes as C
        assert st["n_prev_host"] + n_steps < st["cap"], "decoder state capacity exceeded"
        d = self._desc(st, st["n_prev_host"] if st["lockstep"] else -1, overshoot_weight)
        out = torch.empty(n_steps, st["B"], device=self.device, dtype=torch.int64)
        self.ops.h.check(self.ops.lib.simulst_cif_decode(self.ops.h.ptr, C.byref(d), st["layer_structs"], st["tok"].data_ptr(),
                                                         out.data_ptr(), n_steps, int(mask_eos)), "simulst_cif_decode")
        st["n_prev_host"] += n_steps
        return out

    def stream_steps(self, st, ctl, n_iter: int, overshoot_weight: float = 1.0):
        """n_iter masked policy()/predict() rounds of a batch of streams (simulst_cif_stream_steps)"""
        import ctypes as C
        d = self._desc(st, -1, overshoot_weight)
        self.ops.h.check(self.ops.lib.simulst_cif_stream_steps(self.ops.h.ptr, C.byref(d), st["layer_structs"],
                                                               st["tok"].data_ptr(), C.byref(ctl), n_iter),
                         "simulst_cif_stream_steps")

    def greedy_offline(self, cif_btd: torch.Tensor, cif_len: torch.Tensor, n_steps: int, mask_eos: bool = True,
                       overshoot_weight: float = 1.0, fused: bool = True):
        """Batched greedy decode over the integrated vectors of a finished source (eval/generate.py:187-209 through
        SequenceGenerator, beam 1).  cif_btd [B, n, D] (rows beyond cif_len unused), cif_len [B].  Returns tokens [B, n_steps]."""
        B, n, D = cif_btd.shape
        key = (B, n_steps + 2, n)
        if not hasattr(self, "_offline_states"):
            self._offline_states = {}
        st = self._offline_states.get(key)
        if st is None:
            st = self._offline_states[key] = self.new_device_state(B, cap=key[1], n_cap=n)
        st["n_prev"].zero_()
        st["n_prev_host"], st["lockstep"] = 0, True
        st["cif_len"] = cif_len.to(device=self.device, dtype=torch.int32).contiguous()
        self.project_cif(st, cif_btd.contiguous(), 0, n)
        st["tok"].fill_(self.cfg.eos)
        if fused:
            return self.decode_steps(st, n_steps, mask_eos, overshoot_weight).t().contiguous()
        # per-op launches from the host (reference-shaped control flow; parity tests)
        hst = self.new_state(B, cap=n_steps + 2)
        out = torch.empty(n_steps, B, device=self.device, dtype=torch.int64)
        toks = st["tok"]
        for s_ in range(n_steps):
            logits, overshoot = self.step(hst, toks, cif_btd, st["cif_len"], overshoot_weight)
            toks = self.ops.greedy_argmax(logits, pad_idx=self.cfg.padding_idx, eos_idx=self.cfg.eos,
                                          mask_eos=mask_eos or s_ == 0, eos_bias=overshoot.contiguous(), out=out[s_])
            self.commit(hst)
        return out.t().contiguous()

    def max_positions(self):
        return self.cfg.max_target_positions

    def new_state(self, B: int, cap: int = 128):
        cfg = self.cfg
        H, d = cfg.num_heads, cfg.head_dim
        return {"k": [torch.zeros(B, H, cap, d, device=self.device, dtype=self.dtype) for _ in range(cfg.decoder_layers)],
                "v": [torch.zeros(B, H, cap, d, device=self.device, dtype=self.dtype) for _ in range(cfg.decoder_layers)],
                "n_prev": torch.zeros(B, device=self.device, dtype=torch.int32), "n_prev_host": 0, "cap": cap}

    def step(self, st, last_tokens, cif_btd, cif_lengths, overshoot_weight=1.0):
        """CIFDecoder.forward with incremental_state (:692-724). cif_btd [B,n,D], cif_lengths [B] (host or
        device ints). Returns logits [B,V] fp32 with the EOS overshoot bias applied."""
        ops, cfg, Wd = self.ops, self.cfg, self.w
        B = last_tokens.size(0)
        from .decoder import ensure_positions
        ensure_positions(Wd, st["cap"] + cfg.padding_idx + 2)
        u = st["n_prev_host"] + 1                             # len([eos] + hyp)
        cl = cif_lengths.to(self.device).to(torch.int64)
        idx = (cl.clamp(max=u) - 1).clamp(min=0)
        cif_t = cif_btd[torch.arange(B, device=self.device), idx].contiguous()          # [B,D] gather (:622-628)
        pos_row = (st["n_prev"] + (cfg.padding_idx + 1)).contiguous()
        x = ops.embed_tokens(last_tokens, Wd.E, Wd.pos, pos_row, self.embed_scale)
        for l, L in enumerate(Wd.layers):
            qkv = ops.linear(x, L["wqkv"], L["bqkv"], ln=(L["ln1_g"], L["ln1_b"]))
            ctx = ops.decoder_self_attention(qkv, st["k"][l], st["v"][l], st["n_prev"])
            x = ops.linear(ctx, L["wo"], L["bo"], epilogue=EPI_BIAS_RES, residual=x)
            kk = ops.linear(cif_t, L["c_wk"], L["c_bk"])
            hdn = ops.linear(x, L["c_wq"], None, epilogue=_lib.EPI_BIAS_RES_GELU, residual=kk,
                             ln=(L["ln2_g"], L["ln2_b"]))
            x = ops.linear(hdn, L["c_wo"], L["c_bo"], epilogue=EPI_BIAS_RES, residual=x)
            hdn = ops.linear(x, L["fc1"], L["b1"], epilogue=EPI_BIAS_GELU, ln=(L["ln3_g"], L["ln3_b"]))
            x = ops.linear(hdn, L["fc2"], L["b2"], epilogue=EPI_BIAS_RES, residual=x)
        if cfg.cif_highway:
            y = ops.layernorm(x, Wd.ln_g, Wd.ln_b)
            y = (y.float() + cif_t.float()).to(self.dtype)
            logits = ops.linear(y, Wd.out_proj, None, epilogue=EPI_BIAS_F32OUT)
        else:
            logits = ops.linear(x, Wd.out_proj, None, epilogue=EPI_BIAS_F32OUT, ln=(Wd.ln_g, Wd.ln_b))
        overshoot = (u - cl).clamp(min=0).to(torch.float32) * overshoot_weight       # (:716-722)
        return logits, overshoot

    def commit(self, st):
        st["n_prev"] += 1
        st["n_prev_host"] += 1
        assert st["n_prev_host"] < st["cap"]

    STATE_KEY = "simulst_amd.cif_decoder_state"

    def forward(self, prev_output_tokens, encoder_out=None, incremental_state=None, overshoot_weight=1.0, **unused):
        """CIFDecoder.forward as the CIF agent calls it (agents/cif_agent.py:399-404; models/cif_transformer.py:692-724):
        ``encoder_out`` carries ``cif_out`` [n, B, C] and ``cif_lengths`` [B] accumulated by the agent; state lives in
        the caller's ``incremental_state``.  Returns (logits [B, 1, V] with the EOS overshoot bias applied, {})."""
        if incremental_state is None:
            raise NotImplementedError("simulst_amd.CIFDecoder.forward is the incremental (inference) path")
        B, n_written = prev_output_tokens.size(0), prev_output_tokens.size(1) - 1
        st = incremental_state.get(self.STATE_KEY)
        if st is None or n_written + 2 > st["cap"]:
            new = self.new_state(B, cap=max(64, 2 * (n_written + 8)))
            if st is not None:
                for l in range(self.cfg.decoder_layers):
                    new["k"][l][:, :, :st["cap"]] = st["k"][l]
                    new["v"][l][:, :, :st["cap"]] = st["v"][l]
            st = incremental_state[self.STATE_KEY] = new
        st["n_prev"].fill_(n_written)
        st["n_prev_host"] = n_written
        cif = encoder_out["cif_out"][0].to(device=self.device, dtype=self.dtype).transpose(0, 1).contiguous()
        last = prev_output_tokens[:, -1].to(device=self.device, dtype=torch.int64).contiguous()
        logits, overshoot = self.step(st, last, cif, encoder_out["cif_lengths"][0], overshoot_weight)
        logits = logits.clone()
        logits[:, self.cfg.eos] += overshoot.to(logits.device)
        return logits.unsqueeze(1), {"attn": [None], "inner_states": None}

    __call__ = forward


@register_model("cif_transformer")
class CIFTransformerModel(FairseqModelSurface):
    def __init__(self, cfg: ModelConfig, weights, device="cuda", dtype=torch.float32, ops=None):
        self.cfg = cfg
        self._deferred = None
        self.ops = ops or Ops()
        self.device, self.dtype = torch.device(device), dtype
        self.encoder = CIFEncoder(cfg, weights, device, dtype, self.ops)
        self.decoder = CIFDecoder(cfg, weights, device, dtype, self.ops)

    def get_normalized_probs(self, net_output, log_probs=True):
        lg = net_output[0].float()
        return torch.log_softmax(lg, -1) if log_probs else torch.softmax(lg, -1)

    def max_decoder_positions(self):
        return self.cfg.max_target_positions

    def generate_offline(self, src_tokens, src_lengths, n_steps=None, mask_eos=False, overshoot_weight=1.0, fused=True):
        """task.inference_step with beam 1 for the CIF model (eval/generate.py:200-209; exp/infer_st.yaml:2-5): encoder +
        integrate-and-fire once, then the position-synchronous greedy loop on the device.  Returns tokens [B, n] and the
        encoder dict."""
        enc = self.encoder.forward(src_tokens, src_lengths)
        if n_steps is None:
            n_steps = int(0.1 * src_tokens.size(1) + 10)
        toks = self.decoder.greedy_offline(enc["cif_out_btd"], enc["cif_lengths"][0], n_steps, mask_eos, overshoot_weight,
                                           fused=fused)
        return toks, {"encoder": enc}


class CIFAgent:
    """agents/cif_agent.py: READ while cif_lengths <= len(hyp) and the source has not ended (:385-389),
    otherwise one decoder step and WRITE; cif_out / cif_lengths accumulate across READs (:327-343)."""

    def __init__(self, model: CIFTransformerModel, max_len_a: int = 1, max_len_b: int = 0,
                 overshoot_weight: float = 1.0):
        self.model = model
        enc = model.encoder
        self.stride_ms = enc.conv_layer_stride() * 10
        self.right_context, self.segment_length = enc.right_context, enc.segment_length
        self.max_len = lambda x: max_len_a * x + max_len_b
        self.overshoot_weight = overshoot_weight
        self.eos = model.cfg.eos

    def update_model_encoder(self, states: States):
        src = states.source
        update_len = src.pos - states.last_update_source_len
        if update_len == 0 and states.finish_read():
            return
        finish = (update_len < self.expected_frames) or states.finish_read()
        out = self.model.encoder.infer(src.fbank[:src.pos].unsqueeze(0), torch.tensor([src.pos]),
                                       states.enc_incremental_states, finish=finish)
        if getattr(states, "cif_out", None) is None:
            states.cif_out, states.cif_len = out["cif_out_btd"], int(out["cif_lengths"][0])
        else:
            states.cif_out = torch.cat([states.cif_out, out["cif_out_btd"]], dim=1)
            states.cif_len += int(out["cif_lengths"][0])
        assert states.cif_out.size(1) == states.cif_len, "length mismatch"
        states.last_update_source_len = src.pos

    def run_utterance(self, fbank: torch.Tensor):
        from .latency import average_lagging
        src = FrameSource(fbank)
        states = States(src)
        states.cif_out = None
        dec = self.model.decoder
        dst = dec.new_state(1, cap=int(self.max_len(fbank.size(0))) + 4)
        actions, delays = [], []
        first = (self.segment_length + self.right_context) * self.stride_ms // 10
        nxt = self.segment_length * self.stride_ms // 10
        while True:
            if states.cif_out is None:
                self.expected_frames = first
                read = True
            else:
                read = states.cif_len <= len(states.target) and not states.finish_read()
                if read:
                    self.expected_frames = nxt
            if read:
                actions.append("R")
                if src.finished:
                    raise RuntimeError("READ after source finished")
                src.read(self.expected_frames)
                self.update_model_encoder(states)
                continue
            actions.append("W")
            last = ([self.eos] + states.target)[-1]
            logits, overshoot = dec.step(dst, torch.tensor([last], device=self.model.device), states.cif_out,
                                         torch.tensor([states.cif_len]), self.overshoot_weight)
            tok = int(self.model.ops.greedy_argmax(logits, pad_idx=-1, eos_idx=self.eos,
                                                   eos_bias=overshoot.contiguous())[0].item())
            states.target.append(tok)
            dec.commit(dst)
            delays.append(src.elapsed_ms())
            if tok == self.eos or len(states.target) > self.max_len(src.pos):
                break
        return {"tokens": list(states.target), "delays_ms": delays, "actions": "".join(actions),
                "AL": average_lagging(delays, src.total_ms()), "n_cif": states.cif_len}


class BatchedCIFStreamingAgent(CIFAgent):
    """B simultaneous CIF streams through ONE encoder / decoder batch, every row taking its own READ / WRITE decisions (the
    reference streams one utterance per process: CIFLayer.infer raises for B > 1, models/cif_transformer.py:199-200).  Sources
    advance in lockstep; after each chunk every row writes while it holds more integrated vectors than tokens (the complement
    of the agent's READ condition, agents/cif_agent.py:385-389) -- masked position-synchronous steps on the device
    (simulst_cif_stream_steps).  A row's decisions depend only on its own state and on the source released so far, so its
    tokens, delays and action string are those of ``CIFAgent.run_utterance`` on that utterance alone
    (tests/test_hip_cif_decode.py)."""

    def run_batch(self, fbank: torch.Tensor, self_paced: bool = False, encoder: str = "chunked", lengths=None):
        """fbank [B, T, 80] (equal lengths; ``lengths`` is accepted for interface parity and must not differ between rows: the
        batched CIF integration advances its rows in lockstep).  One record per row, same keys as run_utterance.

        self_paced=True: the evaluation form (agent.BatchedStreamingAgent.run_batch): every chunk is encoded and integrated first,
        then ONE device loop decodes; a row that would READ takes its next chunks inside the commit (this policy's READ does not
        look at the decoder, so it costs no decoder step) -- at most cap rounds.  encoder="offline": the encoder states of one
        offline forward, cut at the rows the streaming schedule releases, go through the same chunk-by-chunk CIF integration."""
        if lengths is not None and len({int(x) for x in lengths} | {fbank.size(1)}) > 1:
            raise ValueError("BatchedCIFStreamingAgent: the rows of a batch must have the same number of frames")
        if self_paced:
            return self._run_batch_self_paced(fbank, encoder)
        if encoder != "chunked":
            raise ValueError("the lockstep form streams through encoder.infer; encoder='offline' needs self_paced=True")
        return self._run_batch_lockstep(fbank)

    def _run_batch_self_paced(self, fbank: torch.Tensor, encoder: str):
        if encoder not in ("chunked", "offline"):
            raise ValueError(f"encoder={encoder!r}: 'chunked' or 'offline'")
        model, dec, enc = self.model, self.model.decoder, self.model.encoder
        cfg, dev = model.cfg, model.device
        B, T = fbank.size(0), fbank.size(1)
        fbank = fbank.to(dev)
        cap = int(self.max_len(T)) + 4
        first = (self.segment_length + self.right_context) * self.stride_ms // 10
        nxt = self.segment_length * self.stride_ms // 10
        positions, pos = [], 0
        while pos < T:
            pos = min(pos + (nxt if positions else first), T)
            positions.append(pos)
        plan_rows = enc.stream_row_schedule(positions)
        n_cap = int((plan_rows[-1] + 1) / enc.cif_layer.beta) + 4
        st = dec.new_device_state(B, cap=cap, n_cap=n_cap)
        st["lockstep"] = False
        cst = enc.cif_layer.new_batched_state(B, cfg.embed_dim, n_cap)
        src = FrameSource(fbank[0])
        off = None
        if encoder == "offline":
            off = S2TEmformerEncoder.forward(enc, fbank, torch.full((B,), T, device=dev))["encoder_out_btd"]
            if off.size(1) != plan_rows[-1]:
                raise RuntimeError(f"offline encoder returned {off.size(1)} rows, the streaming schedule releases {plan_rows[-1]}")
        enc_state, table, ms, mlen, r0 = {}, [], [], [], 0
        for i, pos in enumerate(positions):
            src.read(pos - src.pos)
            finish = i == len(positions) - 1
            if off is None:
                out = S2TEmformerEncoder.infer(enc, fbank[:, :pos], torch.full((B,), pos), enc_state, finish=finish)["encoder_out_btd"]
                if r0 + out.size(1) != plan_rows[i]:
                    raise RuntimeError(f"streaming encoder released {r0 + out.size(1)} rows after chunk {i}, predicted {plan_rows[i]}")
            else:
                out = off[:, r0:plan_rows[i]]
            r0 = plan_rows[i]
            enc.cif_layer.infer_batched(out.contiguous(), cst, finish)
            table.append(cst["cif_len"].clone())
            ms.append(src.elapsed_ms()); mlen.append(int(self.max_len(src.pos)))
        n_chunks = len(positions)
        i32 = dict(device=dev, dtype=torch.int32)
        sched_len = torch.stack(table, 1).contiguous()                            # [B][n_chunks]
        dec.project_cif(st, cst["cif"], 0, min(n_cap, int(sched_len[:, -1].max().item())))
        sched = torch.tensor([[ms] * B, [mlen] * B], **i32)                       # [2][B][n_chunks]
        st["cif_len"] = sched_len[:, 0].clone()
        u8 = dict(device=dev, dtype=torch.uint8)
        online, done = torch.full((B,), 1 if n_chunks > 1 else 0, **u8), torch.zeros(B, **u8)
        hyp = torch.zeros(B, cap, device=dev, dtype=torch.int64)
        delays, tok_chunk, chunk_idx = torch.zeros(B, cap, **i32), torch.zeros(B, cap, **i32), torch.zeros(B, **i32)
        st["tok"].fill_(cfg.eos)
        ctl = _lib.CifStreamCtl(online.data_ptr(), done.data_ptr(), delays.data_ptr(), hyp.data_ptr(), cap, 0, 0, n_chunks,
                                sched_len.data_ptr(), sched[0].data_ptr(), sched[1].data_ptr(), chunk_idx.data_ptr(),
                                st["cif_len"].data_ptr(), tok_chunk.data_ptr(), None)
        bound, first, n_run = mlen[-1] + 2, mlen[-1] + 1, 0
        while True:                                        # every round of an unfinished row is a WRITE: cap + 1 rounds, unless EOS
            n = max(1, min(64, first - n_run) if n_run < first else min(8, bound - n_run))
            dec.stream_steps(st, ctl, n, self.overshoot_weight)
            n_run += n
            if bool(done.all().item()):
                break
            if n_run >= bound:
                raise RuntimeError("self-paced CIF rows unfinished after cap rounds")
        from .agent import self_paced_records
        clen_h = st["cif_len"].tolist()
        return self_paced_records(hyp, delays, tok_chunk, st["n_prev"], chunk_idx, cap, [src.total_ms()] * B, "n_cif",
                                  lambda b, c: clen_h[b])

    def _run_batch_lockstep(self, fbank: torch.Tensor):
        from .latency import average_lagging
        model, dec, enc = self.model, self.model.decoder, self.model.encoder
        cfg, dev = model.cfg, model.device
        B, T = fbank.size(0), fbank.size(1)
        fbank = fbank.to(dev)
        cap = int(self.max_len(T)) + 4
        n_cap = int((T // enc.stride + 2 * self.right_context + 8) / enc.cif_layer.beta) + 4
        st = dec.new_device_state(B, cap=cap, n_cap=n_cap)
        st["lockstep"] = False
        cst = enc.cif_layer.new_batched_state(B, cfg.embed_dim, n_cap)
        st["cif_len"] = cst["cif_len"]
        u8 = dict(device=dev, dtype=torch.uint8)
        online, done = torch.ones(B, **u8), torch.zeros(B, **u8)
        hyp = torch.zeros(B, cap, device=dev, dtype=torch.int64)
        delays = torch.zeros(B, cap, device=dev, dtype=torch.int32)
        st["tok"].fill_(cfg.eos)
        enc_state = {}
        src = FrameSource(fbank[0])
        actions = [[] for _ in range(B)]
        n_written = [0] * B
        cif_len_h = [0] * B
        last_update = 0
        expected = (self.segment_length + self.right_context) * self.stride_ms // 10
        alive = list(range(B))
        while alive:
            # ---- READ phase: every unfinished stream takes the next chunk (agents/cif_agent.py:296-346)
            for b in alive:
                actions[b].append("R")
            if src.finished:
                raise RuntimeError("READ after source finished")
            src.read(expected)
            finish = (src.pos - last_update) < expected or src.finished
            out = S2TEmformerEncoder.infer(enc, fbank[:, :src.pos], torch.full((B,), src.pos), enc_state, finish=finish)
            enc.cif_layer.infer_batched(out["encoder_out_btd"].contiguous(), cst, finish)
            last_update = src.pos
            expected = self.segment_length * self.stride_ms // 10
            new_len = cst["cif_len"].tolist()                      # one read-back per chunk: loop control + projection window
            lo, hi = min(cif_len_h), max(new_len)
            cif_len_h = new_len
            dec.project_cif(st, cst["cif"], lo, hi)
            # ---- WRITE phase: masked steps until no row can write any more
            online.fill_(0 if src.finished else 1)
            ctl = _lib.CifStreamCtl(online.data_ptr(), done.data_ptr(), delays.data_ptr(), hyp.data_ptr(), cap,
                                    src.elapsed_ms(), int(self.max_len(src.pos)), 0, None, None, None, None, None, None, None)
            while True:
                if src.finished:
                    n_iter = 8                                     # rows write until EOS / the length cap
                else:
                    n_iter = max([cif_len_h[b] - n_written[b] for b in alive] + [0])
                if n_iter > 0:
                    dec.stream_steps(st, ctl, n_iter, self.overshoot_weight)
                n_prev = st["n_prev"].tolist()
                done_h = done.tolist()
                for b in alive:
                    actions[b].extend("W" * (n_prev[b] - n_written[b]))
                    n_written[b] = n_prev[b]
                alive = [b for b in alive if not done_h[b]]
                if not src.finished or not alive:
                    break
        hyp_h, delays_h = hyp.tolist(), delays.tolist()
        recs = []
        for b in range(B):
            n = n_written[b]
            d = [int(x) for x in delays_h[b][:n]]
            recs.append({"tokens": hyp_h[b][:n], "delays_ms": d, "actions": "".join(actions[b]),
                         "AL": average_lagging(d, src.total_ms()), "n_cif": cif_len_h[b]})
        return recs


@register_model_architecture("cif_transformer", "cif_transformer_s")
def cif_transformer_s_arch(args):
    """models/cif_transformer.py:727-735."""
    from .model import _default, s2t_emformer_s
    for k, v in (("cif_beta", 1.0), ("cif_sg_alpha", False), ("cif_conv_kernel", 3), ("cif_highway", False)):
        _default(args, k, v)
    if isinstance(args, dict):
        args["ctc_layer"] = True
    else:
        args.ctc_layer = True
    s2t_emformer_s(args)
