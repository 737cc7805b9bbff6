"""Temporal training loops around the hot path, and their multi-GPU form.

Single process: the window loop of the reference's static-temporal and
dynamic-temporal TGCN scripts (benchmarking/static-temporal-tgcn/seastar/
train.py:153-205, benchmarking/dynamic-temporal-tgcn/seastar/train.py:179-231):
every window of ``backprop_every`` consecutive snapshots starts from
``hidden_state = None`` and a fresh ``randn`` input, accumulates the loss, divides
it by ``backprop_every + 1`` (sic), backpropagates through time and
takes one optimizer step.

Multi GPU (new -- the reference has no distributed path): windows
carry no state across each other, so window ``w`` goes to rank ``w mod R``; every
rank holds the full (small) graph and model, and one optimizer step consumes R
windows.  The ONLY collective is one all-reduce (sum, then / R) of the flattened
gradient bucket per optimizer step (RCCL over xGMI when the process group is
"nccl"; ~133 KB for TGCN(32 -> 64), i.e. latency bound, hence a single contiguous
bucket and a single call).  With R = 1 the loop is step-for-step the reference's.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .nn import functional as SF
from .nn.pytorch.temporal.tgcn import TGCN


_FUSED_HEAD = True


def set_fused_head(enabled: bool) -> None:
    """True (default): the training loops below run the model head and the per-step loss as one fused launch
    (nn.functional.tgcn_head); False: ``model(...)`` followed by ``torch.mean((y_out - target) ** 2)``, as the
    reference's script spells it."""
    global _FUSED_HEAD
    _FUSED_HEAD = bool(enabled)


class STGraphTGCN(torch.nn.Module):
    """benchmarking/static-temporal-tgcn/seastar/model.py:6-18."""

    def __init__(self, node_features, num_hidden_units, out_features, tgcn_cls=TGCN):
        super().__init__()
        self.temporal = tgcn_cls(node_features, num_hidden_units)
        self.linear = torch.nn.Linear(num_hidden_units, node_features)
        self.linear2 = torch.nn.Linear(node_features, out_features)

    def forward(self, g, node_feat, edge_weight, hidden_state):
        h = self.temporal(g, node_feat, edge_weight, hidden_state)
        y = F.relu(h)
        y = SF.linear(y, self.linear.weight, self.linear.bias)
        y_out = SF.linear(y, self.linear2.weight, self.linear2.bias)
        return y_out, y, h

    def step_loss(self, g, node_feat, edge_weight, hidden_state, target, cost=None):
        """``forward`` plus the training loop's ``cost = cost + torch.mean((y_out - target) ** 2)``: returns
        (cost, y, h) (``cost`` None: the loss alone).  With the fused head on (``set_fused_head``, default) relu,
        both Linears, the loss and that addition are one launch."""
        if _FUSED_HEAD:
            h = self.temporal(g, node_feat, edge_weight, hidden_state)
            y, _, loss = SF.tgcn_head(h, self.linear.weight, self.linear.bias, self.linear2.weight,
                                      self.linear2.bias, target, cost=cost if torch.is_tensor(cost) else None)
            return loss, y, h
        y_out, y, h = self(g, node_feat, edge_weight, hidden_state)
        loss = torch.mean((y_out - target) ** 2)
        return (loss if not torch.is_tensor(cost) else cost + loss), y, h


class GradBucket:
    """All parameter gradients as views into ONE contiguous fp32 buffer.

    ``zero()`` replaces ``optimizer.zero_grad()`` (which would drop the views);
    ``all_reduce_mean()`` issues the single collective of the data-parallel step.
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.comm_seconds = 0.0
        self.comm_calls = 0

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero(self) -> None:
        self.flat.zero_()

    def check_views(self) -> None:
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + off * self.flat.element_size():
                raise RuntimeError("a parameter's .grad no longer aliases the bucket "
                                   "(use bucket.zero(), not optimizer.zero_grad())")
            off += p.numel()

    def all_reduce_mean(self, world: int, group=None, timed: bool = False) -> None:
        if world <= 1:
            return
        if timed and self.flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(world)
        self.comm_calls += 1
        if timed and self.flat.is_cuda:
            e1.record()
            self._pending = getattr(self, "_pending", [])
            self._pending.append((e0, e1))

    def collect_comm_time(self) -> float:
        """Seconds spent in the timed all-reduces so far (synchronises the device)."""
        pend = getattr(self, "_pending", [])
        if pend:
            torch.cuda.synchronize()
            self.comm_seconds += sum(a.elapsed_time(b) for a, b in pend) * 1e-3
            self._pending = []
        return self.comm_seconds


def num_windows(total_timestamps: int, backprop_every: int) -> int:
    """static-temporal-tgcn/seastar/train.py:137-144."""
    if backprop_every == 0:
        backprop_every = total_timestamps
    return (total_timestamps + backprop_every - 1) // backprop_every


def windows_of_rank(total_timestamps: int, backprop_every: int, rank: int, world: int):
    """[(step, window index or None)]: window w runs on rank w mod R during optimizer step w // R.
    ``None`` marks a padding step where this rank only joins the all-reduce with zero gradients."""
    n = num_windows(total_timestamps, backprop_every)
    steps = (n + world - 1) // world
    out = []
    for s in range(steps):
        w = s * world + rank
        out.append((s, w if w < n else None))
    return out


def window_input(num_nodes: int, feat: int, epoch: int, window: int, device, seed: int = 0) -> torch.Tensor:
    """The fresh ``torch.randn`` input of a window, made a function of (seed, epoch, window) so
    that a run is reproducible for any number of ranks."""
    gen = torch.Generator(device=device)
    gen.manual_seed((seed * 1_000_003 + epoch) * 1_000_003 + window)
    return torch.randn(num_nodes, feat, device=device, generator=gen)


def train_epoch_static(model, graph, edge_weight, targets, backprop_every: int, optimizer,
                       bucket: GradBucket, feat_size: int, epoch: int = 0, rank: int = 0,
                       world: int = 1, group=None, seed: int = 0, timed_comm: bool = False):
    """One epoch of the static-temporal loop; returns the list of this rank's window losses
    (device tensors, no host sync inside the loop)."""
    total = targets.shape[0]
    if backprop_every == 0:
        backprop_every = total
    n = graph.get_num_nodes()
    losses = []
    for _, w in windows_of_rank(total, backprop_every, rank, world):
        bucket.zero()
        if w is not None:
            cost = 0
            hidden = None
            y_hat = window_input(n, feat_size, epoch, w, targets.device, seed)
            for k in range(backprop_every):
                t = w * backprop_every + k
                if t >= total:
                    break
                cost, y_hat, hidden = model.step_loss(graph, y_hat, edge_weight, hidden, targets[t], cost)
            cost = cost / (backprop_every + 1)
            cost.backward()
            losses.append(cost.detach())
        bucket.all_reduce_mean(world, group, timed_comm)
        optimizer.step()
    return losses


class DynamicSTGraphTGCN(torch.nn.Module):
    """benchmarking/dynamic-temporal-tgcn/seastar/model.py:5-21 (link-prediction head)."""

    def __init__(self, node_features, num_hidden_units, tgcn_cls=TGCN):
        super().__init__()
        self.temporal = tgcn_cls(node_features, num_hidden_units)
        self.linear = torch.nn.Linear(num_hidden_units, node_features)

    def forward(self, g, node_feat, edge_weight, hidden_state):
        h = self.temporal(g, node_feat, edge_weight, hidden_state)
        y = F.relu(h)
        y = SF.linear(y, self.linear.weight, self.linear.bias)
        return y, h

    def decode(self, z, edge_label_index):
        return (z[edge_label_index[0]] * z[edge_label_index[1]]).sum(dim=-1)

    def step_loss(self, g, node_feat, edge_weight, hidden_state, edge_label_index, target, cost=None):
        """``forward`` + ``decode`` + the training loop's ``cost = cost + BCEWithLogitsLoss()(...)``: returns
        (cost, y, h) (``cost`` None: the loss alone).  With the fused head on (``set_fused_head``, default) everything
        after the TGCN cell is five launches forward + backward."""
        if _FUSED_HEAD:
            h = self.temporal(g, node_feat, edge_weight, hidden_state)
            y, loss = SF.link_head(h, self.linear.weight, self.linear.bias, edge_label_index, target,
                                   cost=cost if torch.is_tensor(cost) else None)
            return loss, y, h
        y, h = self(g, node_feat, edge_weight, hidden_state)
        out = self.decode(y, edge_label_index).view(-1)
        loss = F.binary_cross_entropy_with_logits(out, target)
        return (loss if not torch.is_tensor(cost) else cost + loss), y, h


def train_epoch_dynamic(model, graph, pos_neg_edges, pos_neg_targets, backprop_every: int, optimizer,
                        bucket: GradBucket, feat_size: int, epoch: int = 0, rank: int = 0, world: int = 1,
                        group=None, seed: int = 0, norm_fn=None, timed_comm: bool = False):
    """One epoch of the dynamic-temporal loop (dynamic-temporal-tgcn/seastar/train.py:179-231):
    per step ``graph.get_graph(t)``, ``norm`` from the snapshot's in-degrees unless already cached
    for that timestamp, un-weighted GCN kernels, link-prediction loss on ``pos_neg_edges[t]``;
    the last timestamp has no prediction target (``t >= T - 1`` stops).  Backward walks the
    snapshots in reverse through the executor's timestamp stack."""
    total = len(pos_neg_edges)
    if backprop_every == 0:
        backprop_every = total
    norm_fn = norm_fn or in_degree_norm
    n = graph.get_num_nodes()
    dev = pos_neg_targets[0].device
    losses = []
    graph.reset_graph()
    for _, w in windows_of_rank(total, backprop_every, rank, world):
        bucket.zero()
        if w is not None:
            cost = 0
            hidden = None
            y_hat = window_input(n, feat_size, epoch, w, dev, seed)
            graph.get_graph(w * backprop_every)
            for k in range(backprop_every):
                t = w * backprop_every + k
                if t >= total - 1:
                    break
                graph.get_graph(t)
                if graph.get_ndata("norm") is None:
                    graph.set_ndata("norm", norm_fn(graph))
                cost, y_hat, hidden = model.step_loss(graph, y_hat, None, hidden, pos_neg_edges[t], pos_neg_targets[t], cost)
            if not isinstance(cost, int):
                cost = cost / (backprop_every + 1)
                cost.backward()
                losses.append(cost.detach())
        bucket.all_reduce_mean(world, group, timed_comm)
        optimizer.step()
    return losses


class CapturedStaticWindow:
    """The compute of one full BPTT window of the static-temporal loop -- bucket.zero,
    ``backprop_every`` model steps, loss, backward through time -- captured ONCE into a HIP graph
    and replayed per window; the gradient all-reduce and the optimizer step follow eagerly.

    Eagerly the window issues ~6 000 small kernels (|V| = 50K: a few microseconds each) and is
    bound by host launch overhead; the graph replays the identical kernel sequence (same kernels,
    same order, same numerics) from device-side descriptors.  Inputs that change per window live
    in static buffers overwritten before each replay: the window's ``randn`` input and its slice of
    ``targets``.  The collective is deliberately NOT captured: one eager RCCL all-reduce and one
    eager (foreach) Adam step per window cost microseconds next to the window itself, and it keeps
    the N > 1 path free of stream-capture constraints on the communicator.
    """

    def __init__(self, model, graph, edge_weight, targets, backprop_every: int, optimizer,
                 bucket: GradBucket, feat_size: int, world: int = 1, group=None, warmup: int = 3):
        self.B = backprop_every
        n = graph.get_num_nodes()
        dev = targets.device
        self.static_y0 = torch.zeros(n, feat_size, device=dev)
        self.static_targets = torch.zeros((backprop_every,) + tuple(targets.shape[1:]), device=dev)
        self.bucket, self.world, self.group, self.optimizer = bucket, world, group, optimizer

        def body():
            bucket.zero()
            cost = 0
            hidden = None
            y_hat = self.static_y0
            for k in range(self.B):
                cost, y_hat, hidden = model.step_loss(graph, y_hat, edge_weight, hidden, self.static_targets[k], cost)
            cost = cost / (self.B + 1)
            cost.backward()
            return cost.detach()

        # Warm up on a side stream (allocator, lazily built per-edge caches, tracing).  The body
        # only writes gradients, so warming up and capturing leave the training state untouched.
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.cost = body()
        bucket.zero()

    def run(self, y0: torch.Tensor, targets_window: torch.Tensor, timed_comm: bool = False) -> torch.Tensor:
        self.static_y0.copy_(y0)
        self.static_targets.copy_(targets_window)
        self.graph.replay()
        self.bucket.all_reduce_mean(self.world, self.group, timed_comm)
        self.optimizer.step()
        return self.cost.clone()


def train_epoch_static_captured(cw: CapturedStaticWindow, model, graph, edge_weight, targets, optimizer,
                                bucket: GradBucket, feat_size: int, epoch: int = 0, rank: int = 0,
                                world: int = 1, group=None, seed: int = 0, timed_comm: bool = False):
    """``train_epoch_static`` with full windows replayed from the captured graph; a trailing short
    window or a padding step (more ranks than windows left) runs eagerly."""
    total = targets.shape[0]
    B = cw.B
    n = graph.get_num_nodes()
    losses = []
    for _, w in windows_of_rank(total, B, rank, world):
        if w is not None and (w + 1) * B <= total:
            y0 = window_input(n, feat_size, epoch, w, targets.device, seed)
            losses.append(cw.run(y0, targets[w * B:(w + 1) * B], timed_comm))
            continue
        bucket.zero()
        if w is not None:
            cost = 0
            hidden = None
            y_hat = window_input(n, feat_size, epoch, w, targets.device, seed)
            for t in range(w * B, min((w + 1) * B, total)):
                cost, y_hat, hidden = model.step_loss(graph, y_hat, edge_weight, hidden, targets[t], cost)
            cost = cost / (B + 1)
            cost.backward()
            losses.append(cost.detach())
        bucket.all_reduce_mean(world, group)
        optimizer.step()
    return losses


def in_degree_norm(graph) -> torch.Tensor:
    """norm = in_deg^-0.5, inf -> 0, computed on the device from the current snapshot."""
    if hasattr(graph, "in_degrees_tensor"):
        deg = graph.in_degrees_tensor().float()
    else:
        f = graph.csr("fwd")
        deg = (f.row_offset[1:] - f.row_offset[:-1]).float()
    norm = torch.pow(deg, -0.5)
    norm[torch.isinf(norm)] = 0
    return norm.unsqueeze(1)
