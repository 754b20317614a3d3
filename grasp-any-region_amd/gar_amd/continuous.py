"""Continuous batching of regions in the decode loop: rows RETIRE and new regions are ADMITTED inside
one running decode loop, instead of every batch paying for its longest caption in every row.

What the reference does per item (demo/gar_with_mask.py:112-122, evaluation/GAR-Bench/inference.py:158-170): one ``generate``
until EOS with ``max_new_tokens=1024``. Captions range from one letter (GAR-Bench VQA) to paragraphs (DLC-Bench); a static
batch of B items runs max(len) steps in all B rows. Here one decode state of ``slots`` rows is kept alive:

* **shared clock, per-row offset.** The captured decode step (one hipGraph per (slots, Smax) bucket) reads ONE position
  counter ``P`` — the cache row every sequence's next key / value goes to — and a per-row ``left_pad`` — the first cache row of
  that sequence; a row's RoPE position is ``P - left_pad`` and its attention covers rows ``left_pad .. P`` (the left-padded
  batch semantics the kernels already serve, HF: position_ids = cumsum(mask) - 1). A region whose prompt is ``n`` rows long
  is therefore admitted at any time by placing its prefilled keys / values at rows ``P - n .. P - 1`` of a free row of the
  cache and setting that row's ``left_pad = P - n``: nothing about the graph changes.
* **prompt phase through a staging state.** A group of queued regions runs the ordinary prompt phase
  (``GARModel.generate_begin``: vision tower, sequence assembly + RoI replay, prefill, first token, with the planner's passes)
  into a staging KV state; admission copies each sequence's rows into its decode row (154 MB per GAR-1B region: ~60 us at
  HBM speed against ~15 ms of prompt phase), its first token into the row's ``cur`` and into the token log.
* **stopping criterion on the device** (``gar_argmax``: ``finished[row]`` latches the step of the row's first eos id), polled
  every ``poll_every`` steps with ONE small device-to-host copy (the latches + the new token columns).
* **re-basing.** ``P`` only grows; when a new row's whole life no longer fits under ``Smax`` the live rows are shifted down
  by the smallest ``left_pad`` among them (a rare bulk copy: every ``horizon`` steps).

Greedy search only, like ``GARModel.generate``; a row's tokens are those of its own single-region ``generate`` (same kernels on
the same data; in f32 identical, in bf16 up to the batch-size dependent kernel selection INTEGRATION.md describes)."""
from __future__ import annotations

from collections import deque
from typing import Dict, List, Optional, Tuple

import torch

from . import hip


def _round_up(x, m):
    return (x + m - 1) // m * m


class _Row:
    __slots__ = ("ticket", "start_col", "left_pad", "n_prompt", "tokens", "fresh")

    def __init__(self, ticket, start_col, left_pad, n_prompt):
        self.ticket, self.start_col, self.left_pad, self.n_prompt = ticket, start_col, left_pad, n_prompt
        self.tokens: List[int] = []
        self.fresh = True            # admitted since the last poll: its first token sits one column in front of the new ones


class ContinuousBatcher:
    """``submit(sample)`` queues one region (a ``generate`` sample with batch dimension 1) and returns its ticket;
    ``pump()`` advances the loop while there is more queued than ``lookahead``; ``flush()`` runs everything to completion.
    Finished captions appear in ``results`` (ticket -> token ids, its own eos included) and ``pop_finished()``."""

    FIRST_SLOT = 2                       # KV-state slots 0 / 1 of the model belong to generate / GenerationPipeline
    STARVE_CYCLES = 8                    # admission cycles a queue head may be passed over before the loop drains for it

    def __init__(self, model, slots: int, max_new_tokens: int = 1024, eos_token_id=None, poll_every: int = 8,
                 admit_min: Optional[int] = None, lookahead: Optional[int] = None, horizon: Optional[int] = None,
                 validate: bool = True, use_graph: bool = True, smax_multiple: int = 256):
        self.model, self.B = model, int(slots)
        # every batcher owns its two KV-state slots (decode / staging) of the model: two batchers — or a batcher and a caller of
        # generate(state_slot=...) — on one model never share a cache, token log, eos latches or graph
        base = getattr(model, "_next_batcher_slot", self.FIRST_SLOT)
        self.DEC_SLOT, self.STAGE_SLOT = base, base + 1
        model._next_batcher_slot = base + 2
        self._head_waits = 0             # cycles the queue's head has been pushed back (prompt longer than the shared clock)
        self.max_new = int(max_new_tokens)
        eos = [] if eos_token_id is None else ([int(e) for e in eos_token_id] if isinstance(eos_token_id, (list, tuple))
                                               else [int(eos_token_id)])
        if len(eos) > model.MAX_EOS_IDS:
            raise hip.GarError(f"{len(eos)} eos ids: the device-side stopping criterion holds {model.MAX_EOS_IDS}")
        self.eos = eos
        self.poll_every = max(1, int(poll_every))
        # smallest group worth a prompt phase of its own while other rows are decoding (the tile GEMMs want rows); a loop that
        # has nothing to decode, or is flushing, admits whatever is queued
        self.admit_min = max(1, self.B // 4) if admit_min is None else max(1, int(admit_min))
        self.lookahead = self.B if lookahead is None else max(0, int(lookahead))
        self.horizon = max(256, 2 * self.max_new) if horizon is None else max(0, int(horizon))
        self.validate, self.use_graph = validate, use_graph
        self.smax_multiple = max(64, _round_up(int(smax_multiple), 64))      # cache rows come in buckets (graph / state reuse)
        self.queue: deque = deque()
        self.rows: List[Optional[_Row]] = [None] * self.B
        self.results: Dict[int, List[int]] = {}
        self._finished_order: List[int] = []
        self._next_ticket = 0
        self.st = self.skey = self.graph = self.out_tokens = None
        self.Smax = 0
        self.P = self.T = None           # host mirrors of the device counters: next cache row / next token column
        self.T_seen = 0                  # token columns < T_seen have been copied to the host
        # what was run (tests, bench.py --eos-mix)
        self.stats = dict(decode_steps=0, prompt_passes=0, admitted=0, retired=0, rebases=0, polls=0, row_steps=0,
                          live_row_steps=0)

    # ---- queue ------------------------------------------------------------------------------------------------------
    def submit(self, sample: dict) -> int:
        if int(sample["input_ids"].shape[0]) != 1:
            raise ValueError("ContinuousBatcher.submit takes one region per sample (input_ids [1, S])")
        t = self._next_ticket
        self._next_ticket += 1
        self.queue.append((t, sample))
        return t

    def pump(self):
        """advance while more than ``lookahead`` regions are waiting (call after submit: keeps the queue bounded)"""
        while len(self.queue) > self.lookahead:
            self._cycle(False)

    def flush(self) -> Dict[int, List[int]]:
        while self.queue or self.n_active:
            self._cycle(True)
        return self.results

    def pop_finished(self) -> List[Tuple[int, List[int]]]:
        out = [(t, self.results[t]) for t in self._finished_order]
        self._finished_order = []
        return out

    @property
    def n_active(self) -> int:
        return sum(r is not None for r in self.rows)

    # ---- one cycle: admit -> decode k steps -> poll / retire -----------------------------------------------------------
    def _cycle(self, flushing: bool):
        m = self.model
        with torch.cuda.device(m.device):
            self._admit(flushing)
            act = [r for r in self.rows if r is not None]
            if not act:
                return
            # never past any row's max_new_tokens: the number of executed steps is then exactly what the captions need
            k = min(self.poll_every, min(self.max_new - (self.T - r.start_col) for r in act))
            for _ in range(max(0, k)):
                if self.graph is not None:
                    self.graph.replay()
                else:
                    m._decode_step(self.st, self.B, self.Smax, self.out_tokens)
            k = max(0, k)
            self.P += k
            self.T += k
            self.stats["decode_steps"] += k
            self.stats["row_steps"] += k * self.B
            self.stats["live_row_steps"] += k * len(act)
            self._poll()

    def _poll(self):
        """ONE device-to-host copy: the eos latches + the token columns produced since the last poll (and the column in front of
        them, where rows admitted in this cycle hold their first token); retire the rows that are done."""
        st = self.st
        lo = self.T_seen - 1
        block = torch.cat([st["finished"].view(self.B, 1).to(torch.int64), self.out_tokens[:, lo:self.T]], 1).tolist()
        self.stats["polls"] += 1
        self.T_seen = self.T
        retire = []
        for r, row in enumerate(self.rows):
            if row is None:
                continue
            fin, toks = block[r][0], block[r][1:]
            row.tokens.extend(toks if row.fresh else toks[1:])
            row.fresh = False
            length = None
            if fin >= 0:
                length = fin - row.start_col + 1
            elif len(row.tokens) >= self.max_new:
                length = self.max_new
            if length is not None:
                self.results[row.ticket] = row.tokens[:length]
                self._finished_order.append(row.ticket)
                retire.append(r)
        if retire:
            idx = torch.tensor(retire, dtype=torch.int64, device=self.model.device)
            # a free row keeps replaying with the batch: hide its keys (one visible cache row) until it is admitted into again
            st["left_pad"].index_fill_(0, idx, self.Smax - 1)
            for r in retire:
                self.rows[r] = None
            self.stats["retired"] += len(retire)

    # ---- admission ----------------------------------------------------------------------------------------------------
    @staticmethod
    def _group_key(sample):
        pv = sample.get("pixel_values")
        return (0 if pv is None else int(pv.shape[0]), tuple(sample.get("video_frame_tokens") or ()),
                bool(sample.get("feature_replay_video")))

    def _admit(self, flushing: bool):
        free = [r for r, row in enumerate(self.rows) if row is None]
        if not free or not self.queue:
            return
        idle = len(free) == self.B
        # a head that does not fit the running clock has waited long enough: admit nothing until the loop is idle — the clock then
        # restarts at ITS length (later, shorter prompts must not keep the rows busy for ever)
        if self._head_waits >= self.STARVE_CYCLES and not idle:
            return
        # a group = the queue's head and what follows it with the same tile count / modality, as many as there are free rows
        key = self._group_key(self.queue[0][1])
        n_match = sum(1 for _, smp in self.queue if self._group_key(smp) == key)
        if not (flushing or idle) and (len(free) < self.admit_min or n_match < min(self.admit_min, len(free))):
            return                       # (gated on the size of the MATCHING group: a prompt phase for one region starves the tile GEMMs)
        group, rest = [], deque()
        while self.queue:
            item = self.queue.popleft()
            if len(group) < len(free) and self._group_key(item[1]) == key:
                group.append(item)
            else:
                rest.append(item)
        self.queue = rest
        head_ticket = group[0][0]
        lens = [int(s["input_ids"].shape[1]) for _, s in group]
        if idle:
            self._new_base(max(lens))
        else:
            # a prompt longer than the shared clock cannot be right-aligned at it yet, and a row's whole life has to fit under Smax
            fits = [n <= self.P for n in lens]
            if self.P + self.max_new > self.Smax:
                if not self._rebase():
                    fits = [False] * len(group)          # pinned by a long-running row: wait for it
                else:
                    fits = [n <= self.P for n in lens]
            keep = [g for g, f in zip(group, fits) if f]
            back = [g for g, f in zip(group, fits) if not f]
            for item in reversed(back):
                self.queue.appendleft(item)
            group = keep
            self._head_waits = self._head_waits + 1 if (back and back[0][0] == head_ticket) else 0
            if not group:
                return
        if idle:
            self._head_waits = 0
        self._prompt_phase_and_admit(group, free)

    def _new_base(self, n_max: int):
        """nothing is decoding: the clock restarts at the longest prompt of the group (and the cache grows if it has to)"""
        m = self.model
        need = _round_up(n_max + self.max_new + self.horizon, self.smax_multiple)
        if self.st is None or self.Smax < n_max + self.max_new:
            self.Smax = need
            self.skey, self.st = m._llm_state(self.B, self.Smax, self.DEC_SLOT)
            self.out_tokens = m._buf(self.skey, "out_tokens", (self.B, self.Smax), torch.int64, zero=True)
            self.graph = None
        st = self.st
        st["eos_ids"].fill_(-1)
        if self.eos:
            st["eos_ids"][:len(self.eos)].copy_(torch.tensor(self.eos, dtype=torch.int64, device=m.device))
        st["finished"].fill_(-1)
        st["done_count"].zero_()
        st["left_pad"].fill_(self.Smax - 1)
        st["cur"].zero_()
        self._set_clock(n_max, 1)
        self.T_seen = 1
        if self.use_graph and self.graph is None and self.max_new > 1:
            self.graph, _ = m._decode_graph(st, self.B, self.Smax, self.out_tokens, self.skey)

    def _set_clock(self, P: int, T: int):
        c = self.st["counters"]
        c[0:1].fill_(P)
        c[1:2].fill_(P + 1)
        c[2:3].fill_(T)
        c[3:4].zero_()
        self.P, self.T = P, T

    def _prompt_phase_and_admit(self, group, free):
        m, st = self.model, self.st
        samples = [s for _, s in group]
        G = len(samples)
        S = max(int(s["input_ids"].shape[1]) for s in samples)
        dev = m.device
        lens = [int(s["input_ids"].shape[1]) for s in samples]
        ragged = any(n != S for n in lens)
        ids = torch.zeros((G, S), dtype=torch.int64, device=dev)
        mask = torch.zeros((G, S), dtype=torch.int64, device=dev) if ragged else None
        for b, s in enumerate(samples):
            ids[b, S - lens[b]:] = s["input_ids"][0].to(dev)
            if ragged:
                mask[b, S - lens[b]:] = 1
        batch = dict(input_ids=ids, bboxes=[s["bboxes"][0] for s in samples] if samples[0].get("bboxes") is not None else None)
        if ragged:
            batch["attention_mask"] = mask
        if samples[0].get("pixel_values") is not None:
            batch["pixel_values"] = torch.cat([s["pixel_values"] for s in samples])
            if samples[0].get("global_mask_values") is not None:
                batch["global_mask_values"] = torch.cat([s["global_mask_values"] for s in samples])
        if samples[0].get("aspect_ratios") is not None:
            batch["aspect_ratios"] = torch.cat([torch.as_tensor(s["aspect_ratios"]) for s in samples])
        if samples[0].get("feature_replay_video"):
            batch.update(feature_replay_video=True, video_frame_tokens=samples[0].get("video_frame_tokens"))
        # the ordinary prompt phase (planner passes, fused kernels) into the staging state; its head evaluates the stopping
        # criterion for the first token too (finished = 0 for a row whose first token is an eos id)
        pend = m.generate_begin(**batch, max_new_tokens=1, eos_token_id=self.eos or None, state_slot=self.STAGE_SLOT,
                                validate=self.validate)
        if not self.validate:
            m._input_flags = pend.input_flags
            m._raise_on_input_flags()
        sg = pend.st
        rows = free[:G]
        P, T = self.P, self.T
        rows_t = torch.tensor(rows, dtype=torch.int64, device=dev)
        offs = [P - n for n in lens]
        # keys / values of sequence i: staging rows S - n .. S - 1 -> decode row rows[i], cache rows P - n .. P - 1
        for i, r in enumerate(rows):
            n = lens[i]
            st["Kc"][:, r, :, P - n:P].copy_(sg["Kc"][:, i, :, S - n:S])
            st["Vc"][:, r, :, P - n:P].copy_(sg["Vc"][:, i, :, S - n:S])
        first = sg["cur"][:G]
        st["cur"].index_copy_(0, rows_t, first)
        st["left_pad"].index_copy_(0, rows_t, torch.tensor(offs, dtype=torch.int32, device=dev))
        self.out_tokens[rows_t, T - 1] = first
        # first token already an eos id: latched at the column it was written to
        fin0 = torch.where(sg["finished"][:G] >= 0, torch.full((G,), T - 1, dtype=torch.int32, device=dev),
                           torch.full((G,), -1, dtype=torch.int32, device=dev))
        st["finished"].index_copy_(0, rows_t, fin0)
        for i, r in enumerate(rows):
            self.rows[r] = _Row(group[i][0], T - 1, offs[i], lens[i])
        self.stats["prompt_passes"] += 1
        self.stats["admitted"] += G

    # ---- re-basing ----------------------------------------------------------------------------------------------------
    def _rebase(self) -> bool:
        """shift every live row down by the smallest left_pad among them (and the token log by the oldest live column), so that
        new rows fit under Smax again. False if a live row already starts at cache row 0 (nothing to gain: wait for it)."""
        live = [(r, row) for r, row in enumerate(self.rows) if row is not None]
        if not live:
            return False
        d = min(row.left_pad for _, row in live)
        if d <= 0:
            return False
        st, P, T = self.st, self.P, self.T
        dT = min(row.start_col for _, row in live)          # >= 0; columns in front of it belong to retired rows
        for r, row in live:
            lo = row.left_pad
            k = st["Kc"][:, r, :, lo:P].clone()
            st["Kc"][:, r, :, lo - d:P - d].copy_(k)
            v = st["Vc"][:, r, :, lo:P].clone()
            st["Vc"][:, r, :, lo - d:P - d].copy_(v)
            del k, v
            row.left_pad -= d
        if dT > 0:
            keep = self.out_tokens[:, dT:T].clone()
            self.out_tokens[:, :T - dT].copy_(keep)
            for _, row in live:
                row.start_col -= dT
        idx = torch.tensor([r for r, _ in live], dtype=torch.int64, device=self.model.device)
        lp = torch.tensor([row.left_pad for _, row in live], dtype=torch.int32, device=self.model.device)
        st["left_pad"].index_copy_(0, idx, lp)
        # live rows are running (their latches are -1); free rows' latches are reset at admission
        self._set_clock(P - d, T - dT)
        self.T_seen = self.T
        self.stats["rebases"] += 1
        return self.P + self.max_new <= self.Smax
