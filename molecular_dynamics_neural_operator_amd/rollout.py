"""On-device autoregressive rollout and ensemble sharding.

`RolloutEngine` drives `mdno_rollout_plan_*` (include/mdno.h): the whole loop of
recursive_propagation (graph_kernel.py:396-413) — graph rebuild on the newest frame, forward,
window slide — stays in HBM; one step is captured into a hipGraph and replayed.

Ensemble members are independent trajectories (block-diagonal graph, no cross-member edges; the
reference's own batching never creates them either, dataset.py:41-45).  Across GPUs they are
sharded by member with NO collective during stepping; `gather_trajectories` is the single
RCCL all-gather (over xGMI on an MI355X node) that collects the finished trajectories.  It
replaces torch_geometric.nn.DataParallel (graph_kernel.py:528), which broadcasts parameters and
gathers outputs every step inside one process.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import MdnoError, STATUS_EDGE_OVERFLOW, check, f32, ptr, raise_on_status, require_gpu


def default_edge_cap(members: int, n_atoms: int, threshold: float, density: float = 0.1, slack: float = 1.6) -> int:
    """Edge capacity for W_e: expected in-degree of a uniform cloud at `density` atoms/A^3 within
    `threshold`, times `slack`, bounded by the complete graph."""
    import math
    deg = 4.0 / 3.0 * math.pi * threshold ** 3 * density + 1.0
    per_member = min(n_atoms * n_atoms, int(n_atoms * deg * slack) + n_atoms)
    return max(members * per_member, members * n_atoms)


# conv_mode "auto", decided on the window an engine is reset with (include/mdno.h mdno_conv_mode_for_graph: factored
# from a mean degree of 40 and 16,384 edges per member on — measured on one box: chains of 600-1,300 atoms at 25-30
# edges per atom are 3-25 % faster materialised, the 504-atom box at 120 per atom 2.2x faster factored).  Protein-like
# chains (10-30 neighbours within 8 A) stay materialised whatever their length; dense boxes go factored.
AUTO_FACTORED_MIN_DEGREE = 40          # (probe bounds only: the rule itself lives in the library)
AUTO_FACTORED_MIN_EDGES = 16384        # per member
AUTO_MATERIALIZED_MAX_WORKSPACE = 64 << 30


class RolloutEngine:
    """Owns the trajectory buffer [W+max_steps, M, N, 3], the workspace and the captured step."""

    def __init__(self, model, members: int, n_atoms: int, window: int, threshold: float = 8.0,
                 max_steps: int = 1000, edge_cap: Optional[int] = None, device=None, use_graph: bool = True):
        self.lib = _lib.load()
        self.device = require_gpu(device if device not in (None, "cuda") else None)
        self.model = model
        self.M, self.N, self.W = int(members), int(n_atoms), int(window)
        self.threshold = float(threshold)
        self.max_steps = int(max_steps)
        self.edge_cap = int(edge_cap) if edge_cap is not None else self.M * self.N * self.N
        self.edge_cap = max(self.edge_cap, self.M * self.N)
        # no capacity given and the complete-graph bound is large (N > 256): the capacity is fitted to the graph of
        # the window the engine is reset with (4x its edges), and the workspace allocated then — W_e at N^2 edges
        # would be 16 GB at N = 1,000 for a chain that has 30,000
        self._fit_cap = edge_cap is None and self.M * self.N * self.N > 65536
        self.pack = model.param_pack(self.device) if hasattr(model, "param_pack") else model
        if not isinstance(self.pack, ops.ParamPack):
            raise MdnoError("model must be a KernelNN (or an ops.ParamPack)")
        dev = self.device
        # what conv_mode "auto" resolves to at this capacity (include/mdno.h MDNO_CONV_AUTO)
        self.conv_mode = {v: k for k, v in _lib.CONV_MODES.items()}[
            int(self.lib.mdno_resolve_conv_mode(self.pack.ref, self.M, self.edge_cap))]
        self.traj = torch.zeros((self.W + self.max_steps, self.M, self.N, 3), dtype=torch.float32, device=dev)
        self.workspace = None
        if not self._fit_cap:
            nbytes = self.lib.mdno_rollout_workspace_bytes(self.pack.ref, self.M, self.N, self.edge_cap)
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.edges_per_step = torch.zeros(self.max_steps, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.aa = None
        self.plan = C.c_void_p()
        # capture needs a real (non-default) stream
        self.stream = torch.cuda.Stream(device=dev)
        self.use_graph = bool(use_graph)
        self.steps_done = 0
        self._sample_first_step = False
        self._timer_records = 0        # capacity of the attached per-kernel timer (0 = none): re-attached to a rebuilt plan
        self.regrown = []              # (first re-run step, old edge_cap, new edge_cap) per growth of a fitted capacity
        self.mode_changes = []         # (step, from, to): "auto" changed formulation when the capacity grew mid-trajectory
        # "auto": what the edge capacity suggests until a window is known; reset() decides on its graph
        self._auto = (getattr(model, "conv_mode", None) if hasattr(model, "param_pack") else self.pack.conv_mode) == "auto"

    def _pack_for(self, conv_mode: str):
        if hasattr(self.model, "param_pack"):
            return self.model.param_pack(self.device, conv_mode=conv_mode)
        p = self.pack
        return ops.ParamPack({v: p.tensors[k] for k, v in ops.ParamPack.KEYS.items() if k in p.tensors},
                             p.struct.depth, self.device, p.gemm_mode, conv_mode)

    def _probe_edges(self) -> Tuple[int, bool]:
        """Edges of the radius graph of the window's last frame, counted up to the bounds the decisions below look
        at; (count, True) when there are more."""
        R = self.M * self.N
        probe_cap = max(AUTO_FACTORED_MIN_DEGREE * R, self.M * AUTO_FACTORED_MIN_EDGES) + R
        g = ops.radius_graph(self.traj[self.W - 1].reshape(R, 3), self.N, self.threshold, edge_cap=probe_cap)
        return int(g.num_edges.item()), bool(int(g.status.item()) & STATUS_EDGE_OVERFLOW)

    def _fit_capacity(self, e: int, over: bool) -> bool:
        R = self.M * self.N
        cap = self.M * self.N * self.N if over else min(self.M * self.N * self.N, max(4 * e, e + 16 * R) + R)
        if cap == self.edge_cap:
            return False
        self.edge_cap = cap
        return True

    def _wanted_mode(self, e: int, over: bool) -> str:
        """conv_mode "auto" on a counted graph of this engine's members (AUTO_FACTORED_* above)."""
        if over:
            e = max(e, self.M * max(AUTO_FACTORED_MIN_DEGREE * self.N, AUTO_FACTORED_MIN_EDGES))
        return {v: k for k, v in _lib.CONV_MODES.items()}[
            int(self.lib.mdno_conv_mode_for_graph(self.pack.ref, self.M, self.N, int(e)))]

    def _apply_mode(self, want: str, record_at: Optional[int] = None) -> bool:
        """Switch to formulation `want` where the model's dimensions (and, for materialized, the workspace bound) allow
        it.  Returns True if the engine changed formulation (its plan must then be rebuilt).  `record_at`: the step
        from which the new formulation applies when the change happens MID-trajectory (a regrown capacity) — recorded
        in `mode_changes`, step 0 included; None at reset() (the choice for the whole trajectory is no change)."""
        if want == self.conv_mode:
            return False
        pack = self._pack_for(want)
        if {v: k for k, v in _lib.CONV_MODES.items()}[
                int(self.lib.mdno_resolve_conv_mode(pack.ref, self.M, self.edge_cap))] != want:
            return False                 # (factored is not available for this model's dimensions)
        need = self.lib.mdno_rollout_workspace_bytes(pack.ref, self.M, self.N, self.edge_cap)
        if want == "materialized" and need > AUTO_MATERIALIZED_MAX_WORKSPACE:
            return False                 # W_e at this edge capacity does not fit: stay factored
        if record_at is not None:
            self.mode_changes.append((int(record_at), self.conv_mode, want))
        self.pack, self.conv_mode = pack, want
        return True

    def _resolve_auto(self, e: int, over: bool, record_at: Optional[int] = None) -> bool:
        return self._apply_mode(self._wanted_mode(e, over), record_at)

    def _create_plan(self):
        if self.plan:
            check(self.lib.mdno_rollout_plan_destroy(self.plan), "mdno_rollout_plan_destroy")
            self.plan = C.c_void_p()
        aa_pm = int(self.aa.numel() == self.M * self.N and self.M > 1)
        check(self.lib.mdno_rollout_plan_create(
            C.byref(self.plan), self.pack.ref, ptr(self.traj), self.M, self.W, self.N, self.max_steps,
            ptr(self.aa), aa_pm, self.threshold, self.edge_cap, ptr(self.workspace),
            self.workspace.numel(),
            ptr(self.edges_per_step), ptr(self.status), int(self.use_graph), self.stream.cuda_stream),
            "mdno_rollout_plan_create")
        if self._timer_records:          # the timer lived in the plan just destroyed: its records are gone, the attachment is not
            check(self.lib.mdno_rollout_plan_timer_attach(self.plan, self._timer_records), "timer_attach")

    @property
    def steps_per_launch(self) -> int:
        """0 = plain launches, 1 = one captured step per graph launch, 8 = eight (short chains); include/mdno.h."""
        return int(self.lib.mdno_rollout_plan_steps_per_launch(self.plan)) if self.plan else 0

    def reset(self, window: torch.Tensor, x_aminoacid: torch.Tensor, _conv_mode: Optional[str] = None) -> None:
        """window: f32 [W,M,N,3] (time-major), [W,N,3] for M=1; x_aminoacid i64 [N] or [M*N].
        (_conv_mode: what "auto" resolved to for the whole shard this engine is a group of — GroupedRolloutEngine.)"""
        w = f32(window.to(self.device))
        if w.dim() == 3:
            w = w.unsqueeze(1)
        if tuple(w.shape) != (self.W, self.M, self.N, 3):
            raise MdnoError(f"window shape {tuple(w.shape)} != {(self.W, self.M, self.N, 3)}")
        aa = x_aminoacid.to(device=self.device, dtype=torch.long).contiguous()
        if aa.numel() not in (self.N, self.M * self.N):
            raise MdnoError(f"x_aminoacid has {aa.numel()} entries, expected {self.N} or {self.M * self.N}")
        torch.cuda.current_stream(self.device).synchronize()
        self.stream.synchronize()
        self.traj[:self.W].copy_(w)
        self.status.zero_()
        new_aa = self.aa is None or self.aa.shape != aa.shape
        if new_aa:
            self.aa = aa
        else:
            self.aa.copy_(aa)
        torch.cuda.current_stream(self.device).synchronize()
        changed = False
        e, over = 0, False
        if self._fit_cap or (self._auto and not _conv_mode):
            e, over = self._probe_edges()
        if self._fit_cap:
            changed |= self._fit_capacity(e, over)
        if self._auto:
            changed |= self._apply_mode(_conv_mode) if _conv_mode else self._resolve_auto(e, over)
        need = self.lib.mdno_rollout_workspace_bytes(self.pack.ref, self.M, self.N, self.edge_cap)
        if self.workspace is None or need > self.workspace.numel() or 2 * need < self.workspace.numel():
            self.workspace = None        # (release before allocating: the two could be tens of GB each)
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            changed = True
        if new_aa or changed or not self.plan:
            self._create_plan()
        self.steps_done = 0
        self._sample_first_step = False
        self.regrown, self.mode_changes = [], []

    def first_step_from_sample(self, edge_index: torch.Tensor, edge_attr: torch.Tensor) -> None:
        """Produce frame W from the start sample's OWN graph and edge attributes, as the reference's
        first loop iteration does (graph_kernel.py:401-404: `dataset[start]` carries the graph of the
        window's FIRST frame, dataset.py:189-201; only later steps rebuild it on the newest frame).
        Single-member engines only; call after reset()."""
        if self.M != 1 or self.steps_done != 0:
            raise MdnoError("first_step_from_sample: needs M == 1 and a freshly reset engine")
        graph = ops.coo_to_csr(edge_index.to(self.device), self.N)
        # an explicit edge list runs in the engine's own formulation (the factored form takes any destination-sorted graph)
        out, _ = ops.kernelnn_forward(self.pack, self.traj[:self.W], self.aa, graph,
                                      edge_attr=edge_attr.to(self.device))
        self.traj[self.W, 0].copy_(out)
        self.edges_per_step[0] = int(edge_index.shape[1])
        torch.cuda.current_stream(self.device).synchronize()
        self.steps_done = 1
        self._sample_first_step = True

    def step(self, steps: int) -> None:
        """Enqueue `steps` more frames on the engine's stream (asynchronous)."""
        if not self.plan:
            raise MdnoError("RolloutEngine.reset(window, x_aminoacid) must be called first")
        check(self.lib.mdno_rollout_plan_run(self.plan, self.steps_done, int(steps), self.stream.cuda_stream),
              "mdno_rollout_plan_run")
        self.steps_done += int(steps)

    KERNEL_IDS = {"nnconv": 0, "edge_mlp_gemm1": 1, "edge_mlp_gemm2": 2, "edge_mlp_l0": 3, "radius_graph": 4,
                  "node_prologue": 5, "fc_out": 6, "nnconv_combine": 7, "factored_y": 8}

    def attach_timer(self, max_records: int) -> None:
        """Per-kernel HIP-event timing (measurement aid): subsequent step() calls issue plain
        launches bracketed by events on the engine's stream."""
        check(self.lib.mdno_rollout_plan_timer_attach(self.plan, int(max_records)), "timer_attach")
        self._timer_records = int(max_records)

    def read_timer(self) -> dict:
        """{kernel: (total_ms, launches)}; synchronises the engine's stream first."""
        self.stream.synchronize()
        out = {}
        for name, kid in self.KERNEL_IDS.items():
            ms, n = C.c_double(0.0), C.c_int64(0)
            check(self.lib.mdno_rollout_plan_timer_read(self.plan, kid, C.byref(ms), C.byref(n)), "timer_read")
            out[name] = (ms.value, n.value)
        return out

    def detach_timer(self) -> None:
        self.stream.synchronize()
        check(self.lib.mdno_rollout_plan_timer_detach(self.plan), "timer_detach")
        self._timer_records = 0

    FALLBACK_KEYS = ops.FALLBACK_KEYS

    def fallback_counts(self) -> dict:
        """Which path the products of gemm_mode "split_f16" took since the last `step()` call (include/mdno.h,
        mdno_rollout_plan_fallback_counts): K1 workgroups of the factored conv rerun on bf16 planes, destinations whose
        operands were taken unscaled, edge-MLP products (GEMM x chunk) redone on bf16 planes.  All zero = every product
        ran on two fp16 planes (the fast path); non-zero = same results to fp32 rounding, more matrix work.
        Synchronises the engine's stream."""
        if not self.plan:
            return {k: 0 for k in self.FALLBACK_KEYS}
        out = (C.c_int64 * 4)()
        check(self.lib.mdno_rollout_plan_fallback_counts(self.plan, out, self.stream.cuda_stream), "fallback_counts")
        return {k: int(out[i]) for i, k in enumerate(self.FALLBACK_KEYS)}

    def _grow_and_rerun(self, st: int) -> int:
        """The radius graph of some step outgrew a capacity that was FITTED to the start window (no `edge_cap` given,
        N > 256): the reference keeps building the denser graph (graph_kernel.py:363-368), so the engine does too —
        capacity x4 (at most the complete graph), new workspace and plan, and the steps from the first truncated one
        on are run again (frames before it are untouched: their graphs fitted).  Recorded in `self.regrown`; with
        conv_mode "auto" the larger graph can take the other formulation from that step on (`self.mode_changes`; the
        two agree to fp32 reassociation, not bitwise).  An attached timer is re-attached to the new plan (its records
        up to here are lost with the old one).  Returns the new status word."""
        target = self.steps_done
        eps = self.edges_per_step[:target].cpu()
        hit = (eps >= self.edge_cap).nonzero()
        first = int(hit[0]) if hit.numel() else 0
        if self._sample_first_step:
            first = max(first, 1)                # step 0 ran on the sample's own edge list (cannot overflow)
        old = self.edge_cap
        self.edge_cap = min(self.M * self.N * self.N, 4 * old)
        self.regrown.append((first, old, self.edge_cap))
        if self._auto:
            self._resolve_auto(old, False, record_at=first)       # the graph now has at least `old` edges
        need = self.lib.mdno_rollout_workspace_bytes(self.pack.ref, self.M, self.N, self.edge_cap)
        self.workspace = None
        self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        self.status.zero_()
        torch.cuda.current_stream(self.device).synchronize()
        self._create_plan()
        self.steps_done = first
        self.step(target - first)
        self.stream.synchronize()
        return (st & ~STATUS_EDGE_OVERFLOW) | int(self.status.item())

    def synchronize(self) -> None:
        self.stream.synchronize()
        st = int(self.status.item())
        while (st & STATUS_EDGE_OVERFLOW) and self._fit_cap and self.edge_cap < self.M * self.N * self.N:
            st = self._grow_and_rerun(st)
        if st & STATUS_EDGE_OVERFLOW:
            raise MdnoError(f"radius graph exceeded edge_cap={self.edge_cap}; construct the engine with a larger cap")
        raise_on_status(st, "rollout")

    def run(self, window: torch.Tensor, x_aminoacid: torch.Tensor, steps: int) -> torch.Tensor:
        """reset + step + synchronize; returns the produced frames f32 [steps, M, N, 3] (a view)."""
        self.reset(window, x_aminoacid)
        self.step(steps)
        self.synchronize()
        return self.frames()

    def frames(self) -> torch.Tensor:
        return self.traj[self.W:self.W + self.steps_done]

    def produced(self, first_step: int, steps: int) -> torch.Tensor:
        """Frames of steps first_step .. first_step + steps - 1: f32 [steps, M, N, 3] (a view of the trajectory buffer)."""
        return self.traj[self.W + first_step:self.W + first_step + steps]

    def close(self) -> None:
        if self.plan:
            self.stream.synchronize()
            self.lib.mdno_rollout_plan_destroy(self.plan)
            self.plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupedRolloutEngine:
    """The members of a shard as `groups` independent `RolloutEngine`s (contiguous member ranges), each with its own
    stream and its own captured step, stepped together.  Members never interact and conv_mode "auto" is resolved ONCE,
    on the graph of the whole shard, and given to every group (a group deciding on its own members could take the other
    formulation when densities differ), so the frames are those of one engine holding them all: bitwise for members of
    like magnitude (the ensemble case: perturbed copies of one system), and to fp32 rounding in general — the edge-MLP
    chooses between its fp16 and bf16 plane products per launch from the magnitudes it sees (split_layout.h), and a
    group can see other magnitudes than the whole shard.  A group whose FITTED capacity is outgrown mid-trajectory regrows
    on its own and re-resolves "auto" on its own members (RolloutEngine._grow_and_rerun): from that step on the groups
    may run different formulations — `conv_mode` then reads "mixed:..", every group's `mode_changes` says where — and
    the one-engine equivalence holds to fp32 reassociation only.  What changes is the schedule: one group's edge-MLP (matrix-pipe-bound) and launch tails run
    beside another group's convs (fabric-bound) — 5 % at 8 x 504 atoms, 3 % at 64 (EXPERIMENTS.md section 0.2b).  Same
    reset / step / synchronize / run / frames interface; `traj` and `edges_per_step` are assembled on access."""

    def __init__(self, model, members: int, n_atoms: int, window: int, threshold: float = 8.0, max_steps: int = 1000,
                 edge_cap: Optional[int] = None, device=None, use_graph: bool = True, groups: int = 2):
        self.M, self.N, self.W = int(members), int(n_atoms), int(window)
        g = max(1, min(int(groups), self.M))
        base, extra = divmod(self.M, g)
        self.bounds, lo = [], 0
        for i in range(g):
            hi = lo + base + (1 if i < extra else 0)
            self.bounds.append((lo, hi))
            lo = hi
        self.engines = [RolloutEngine(model, hi - lo, n_atoms, window, threshold, max_steps=max_steps,
                                      edge_cap=None if edge_cap is None else max(1, -(-int(edge_cap) * (hi - lo) // self.M)),
                                      device=device, use_graph=use_graph) for lo, hi in self.bounds]
        self.device = self.engines[0].device
        self.max_steps = int(max_steps)

    @property
    def conv_mode(self):
        modes = {e.conv_mode for e in self.engines}
        return self.engines[0].conv_mode if len(modes) == 1 else "mixed:" + ",".join(sorted(modes))

    @property
    def steps_done(self):
        return self.engines[0].steps_done

    @property
    def regrown(self):
        return [(g,) + r for g, e in enumerate(self.engines) for r in e.regrown]

    @property
    def workspace_bytes(self) -> int:
        return sum(e.workspace.numel() for e in self.engines if e.workspace is not None)

    def reset(self, window: torch.Tensor, x_aminoacid: torch.Tensor) -> None:
        """window f32 [W,M,N,3]; x_aminoacid i64 [N] (shared) or [M*N]."""
        if window.dim() == 3:
            window = window.unsqueeze(1)
        if tuple(window.shape) != (self.W, self.M, self.N, 3):
            raise MdnoError(f"window shape {tuple(window.shape)} != {(self.W, self.M, self.N, 3)}")
        if x_aminoacid.numel() not in (self.N, self.M * self.N):
            raise MdnoError(f"x_aminoacid has {x_aminoacid.numel()} entries, expected {self.N} or {self.M * self.N}")
        per_member = x_aminoacid.numel() == self.M * self.N and self.M > 1
        want = None
        if any(e._auto for e in self.engines):      # the rule one engine holding every member would apply
            e0 = self.engines[0]
            R = self.M * self.N
            probe_cap = max(AUTO_FACTORED_MIN_DEGREE * R, self.M * AUTO_FACTORED_MIN_EDGES) + R
            last = f32(window[self.W - 1].to(self.device)).reshape(R, 3)
            g = ops.radius_graph(last, self.N, e0.threshold, edge_cap=probe_cap)
            n_e, over = int(g.num_edges.item()), bool(int(g.status.item()) & STATUS_EDGE_OVERFLOW)
            if over:
                n_e = max(n_e, self.M * max(AUTO_FACTORED_MIN_DEGREE * self.N, AUTO_FACTORED_MIN_EDGES))
            want = {v: k for k, v in _lib.CONV_MODES.items()}[
                int(e0.lib.mdno_conv_mode_for_graph(e0.pack.ref, self.M, self.N, n_e))]
        for e, (lo, hi) in zip(self.engines, self.bounds):
            e.reset(window[:, lo:hi].contiguous(), x_aminoacid[lo * self.N:hi * self.N] if per_member else x_aminoacid,
                    _conv_mode=want)

    def step(self, steps: int) -> None:
        for e in self.engines:      # asynchronous on each engine's own stream
            e.step(steps)

    def fallback_counts(self) -> dict:
        """Sum of the groups' counters (RolloutEngine.fallback_counts)."""
        tot = {k: 0 for k in RolloutEngine.FALLBACK_KEYS}
        for e in self.engines:
            for k, v in e.fallback_counts().items():
                tot[k] += v
        return tot

    def wait(self) -> None:
        """Block until every group's stream has drained (no status check: `synchronize` does that)."""
        for e in self.engines:
            e.stream.synchronize()

    def synchronize(self) -> None:
        for e in self.engines:
            e.synchronize()

    def run(self, window: torch.Tensor, x_aminoacid: torch.Tensor, steps: int) -> torch.Tensor:
        self.reset(window, x_aminoacid)
        self.step(steps)
        self.synchronize()
        return self.frames()

    @property
    def traj(self) -> torch.Tensor:
        return torch.cat([e.traj for e in self.engines], dim=1)

    def frames(self) -> torch.Tensor:
        return torch.cat([e.frames() for e in self.engines], dim=1)

    def produced(self, first_step: int, steps: int) -> torch.Tensor:
        """Frames of steps first_step .. first_step + steps - 1 of every group, members in order: [steps, M, N, 3] (only
        these frames are copied — `traj` concatenates the groups' whole buffers)."""
        return torch.cat([e.produced(first_step, steps) for e in self.engines], dim=1)

    @property
    def edges_per_step(self) -> torch.Tensor:
        return torch.stack([e.edges_per_step for e in self.engines]).sum(0, dtype=torch.int32)

    def close(self) -> None:
        for e in self.engines:
            e.close()


# --------------------------------------------------------------------------- ensemble sharding
def shard_members(total_members: int, rank: int, world_size: int) -> List[int]:
    """Member m runs on rank m mod world_size."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, total_members, world_size))


def gather_trajectories(local: torch.Tensor, total_members: int, group=None) -> torch.Tensor:
    """All-gather member trajectories: local f32 [T, M_local, N, 3] -> [T, total_members, N, 3] on
    every rank, members in global order.  One collective (RCCL on GPUs; gloo for CPU tests).
    Ranks may hold unequal member counts (total not divisible by world): shards are padded to the
    largest count for the collective and trimmed afterwards."""
    import torch.distributed as dist
    if not dist.is_initialized():
        if local.shape[1] != total_members:
            raise ValueError("not distributed: local shard must hold every member")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [len(shard_members(total_members, r, world)) for r in range(world)]
    if local.shape[1] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[1]} members, expected {counts[rank]}")
    m_max = max(counts)
    T, _, N, D = local.shape
    send = local
    if counts[rank] < m_max:
        pad = torch.zeros((T, m_max - counts[rank], N, D), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=1)
    send = send.contiguous()
    staged = local.is_cuda and dist.get_backend(group) == "gloo"     # CPU rehearsal backend: stage through host
    if staged:
        send = send.cpu()
    recv = torch.empty((world, T, m_max, N, D), dtype=local.dtype, device=send.device)
    dist.all_gather_into_tensor(recv.view(world * T, m_max, N, D), send, group=group)   # rank r's shard = recv[r]
    if staged:
        recv = recv.to(local.device)
    out = torch.empty((T, total_members, N, D), dtype=local.dtype, device=local.device)
    if total_members == world * m_max:      # even shards: member m = j * world + r  ->  one strided copy
        out.view(T, m_max, world, N, D).copy_(recv.permute(1, 2, 0, 3, 4))
    else:                                   # rank r's members are r, r + world, ..: a strided slice each (no index tensors)
        for r in range(world):
            if counts[r]:
                out[:, r::world] = recv[r, :, :counts[r]]
    return out
