"""One training step of the reference (Trainer.train, lib/core/base.py:122-153): forward in training mode, the weighted losses,
backward, Adam - every kernel from libgator_hip.so.  Data parallel: each rank steps on its shard and the flat gradient is
all-reduced (averaged) once per step before the update."""
import torch

from . import losses as L
from . import model as M
from . import ops
from .optim import Adam, FlatParams


class Trainer:
    def __init__(self, state_dict, consts, faces, j_regressor, device='cuda', lr=1e-3, rates=None, seed=0, edge_loss_start=15, dist=None):
        self.params = FlatParams(state_dict, device)
        self.consts = consts
        self.losses = L.MeshLosses(faces, j_regressor, device)
        self.optim = Adam(self.params, lr=lr)
        self.rates = rates if rates is not None else M.Rates()
        self.gen = ops.Generator(seed)
        self.edge_loss_start = edge_loss_start
        self.dist = dist
        self.epoch = 0
        self._graph = None
        self._graphs = {}

    @classmethod
    def from_module(cls, module, faces, j_regressor, device='cuda', **kw):
        """Train the parameters of a gator_amd.models.GATOR module (reference state_dict layout)."""
        return cls({k: v.detach().cpu() for k, v in module.state_dict().items()}, M.consts_from_module(module, device), faces, j_regressor, device, **kw)

    def state_dict(self):
        """Reference-layout state_dict of the trained weights: module.load_state_dict(trainer.state_dict()) puts them on the
        inference kernels."""
        return self.params.state_dict()

    def with_edge(self):
        """EdgeLengthLoss joins the sum from the epoch after `edge_loss_start` on (lib/core/base.py:141-143)."""
        return self.epoch > self.edge_loss_start

    def loss_and_grad(self, pose2d, targets, training=True, with_edge=None):
        P = self.params.views()
        mesh, pose3d = M.gator_forward(P, self.consts, pose2d, self.gen, self.rates, training, self.params.buffers)
        loss, parts = self.losses.total(mesh, pose3d, targets, with_edge=self.with_edge() if with_edge is None else with_edge)
        one = ops.raw_unary(ops.U_AFFINE, loss.detach(), 0.0, 1.0)          # d loss / d loss (autograd's default would be an aten fill)
        grad, = torch.autograd.grad(loss, self.params.flat, grad_outputs=one)
        return loss.detach(), {k: v.detach() for k, v in parts.items()}, grad

    def capture(self, pose2d, targets):
        """Capture forward + losses + backward of one step (this batch shape) into a hipGraph: the ~700 kernel launches of a step
        become one graph launch, the host only replays.  Dropout offsets and Adam's step index come from a device counter that the
        graph itself advances, so every replay draws new masks.  The gradient all-reduce and the Adam launch stay outside the graph.

        The set of losses is part of the captured graph, so one graph exists per value of `with_edge()`: the variant of the current
        epoch is captured here, the other one by the first `step()` that needs it (on the same static input buffers).  Capturing
        does not disturb the training state: the step counter continues from `optim.step_count` (capture after eager steps or after
        `optim.load_state_dict`), and the BatchNorm running statistics that the warm-up passes touch are put back."""
        dev = self.params.flat.device
        if self.gen.counter is None:
            self.gen.device_steps(dev)
        self.optim.device_step = self.gen.counter
        self._x = pose2d.clone()
        self._tg = {k: v.clone() for k, v in targets.items()}
        self._graphs = {}
        self._capture_variant(self.with_edge())
        return self

    def _capture_variant(self, with_edge):
        dev = self.params.flat.device
        saved = {k: v.clone() for k, v in self.params.buffers.items()}
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                        # warm-up off the capture: workspaces and the allocator's pools
            for _ in range(2):
                self._body(with_edge)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        for k, v in saved.items():                           # the warm-up is not a training step: running_mean / running_var /
            self.params.buffers[k].copy_(v)                  # num_batches_tracked back to what the eager trainer would hold
        self.gen.counter.fill_(self.optim.step_count)        # replay n + 1 is step optim.step_count + n + 1 (Adam bias correction, masks)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._body(with_edge)
        self._graphs[bool(with_edge)] = (graph, out)
        self._graph = graph

    def _body(self, with_edge):
        self.gen.begin_step()
        return self.loss_and_grad(self._x, self._tg, with_edge=with_edge)

    def _check_captured_batch(self, pose2d, targets):
        if tuple(pose2d.shape) != tuple(self._x.shape):
            raise ValueError('captured step expects pose2d %s, got %s: call capture() again for a new batch shape' % (tuple(self._x.shape), tuple(pose2d.shape)))
        if set(targets) != set(self._tg):
            raise ValueError('captured step expects target keys %s, got %s' % (sorted(self._tg), sorted(targets)))
        for k, v in targets.items():
            if tuple(v.shape) != tuple(self._tg[k].shape):
                raise ValueError('captured step expects targets[%r] %s, got %s' % (k, tuple(self._tg[k].shape), tuple(v.shape)))

    def step(self, pose2d, targets):
        """optimizer.zero_grad(); loss.backward(); optimizer.step()  (base.py:151-153)"""
        if self._graph is not None:
            self._check_captured_batch(pose2d, targets)
            we = bool(self.with_edge())
            if we not in self._graphs:                       # the epoch crossed edge_loss_start since capture(): second graph
                self._capture_variant(we)
            ops.raw_unary(ops.U_AFFINE, pose2d, 1.0, 0.0, out=self._x)
            for k, v in targets.items():
                ops.raw_unary(ops.U_AFFINE, v, 1.0, 0.0, out=self._tg[k])
            graph, out = self._graphs[we]
            graph.replay()
            loss, parts, grad = out
        else:
            self.gen.begin_step()
            loss, parts, grad = self.loss_and_grad(pose2d, targets)
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            self.dist.all_reduce(grad)
            grad = ops.raw_unary(ops.U_AFFINE, grad, 1.0 / self.dist.get_world_size(), 0.0)
        self.optim.epoch = self.epoch
        self.optim.step(grad)
        return loss, parts


class LiftTrainer:
    """LiftTrainer.train (lib/core/base.py:260-300): the pose lifter alone - GAT forward in training mode, CoordLoss on the lifted
    joints against `cam_joint` with `joint_valid`, backward, Adam.  Takes a gator_amd.models.GAT module (or any reference-layout
    state_dict whose keys carry no `pose_lifter.` prefix)."""

    def __init__(self, state_dict, consts, device='cuda', lr=1e-3, rates=None, seed=0):
        self.params = FlatParams(state_dict, device)
        self.consts = consts
        self.optim = Adam(self.params, lr=lr)
        self.rates = rates if rates is not None else M.Rates()
        self.gen = ops.Generator(seed)
        self.losses = L.MeshLosses([[0, 1, 2]], [[0.0]], device, num_verts=3)        # only CoordLoss is used (base.py:265)
        self.epoch = 0

    @classmethod
    def from_module(cls, module, device='cuda', **kw):
        sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
        J = module.num_joint
        z3 = torch.zeros(1, 3)
        c = M.Consts(J, sd['graph_adj'].numpy(), module.spatial_pos, module.edge_input, z3.numpy(), z3.numpy(), [0], False, device)
        return cls(sd, c, device, **kw)

    def state_dict(self):
        return self.params.state_dict()

    def loss_and_grad(self, img_joint, cam_joint, joint_valid, training=True):
        P = self.params.views()
        B, J = img_joint.shape[0], self.consts.J
        x_out, _ = M.gat_forward(P, self.consts, img_joint.reshape(B, J, 2), self.gen, self.rates, training, p='')
        loss = self.losses.coord(x_out.reshape(B, J, 3), cam_joint, joint_valid)
        one = ops.raw_unary(ops.U_AFFINE, loss.detach(), 0.0, 1.0)
        grad, = torch.autograd.grad(loss, self.params.flat, grad_outputs=one)
        return loss.detach(), grad

    def step(self, img_joint, cam_joint, joint_valid):
        self.gen.begin_step()
        loss, grad = self.loss_and_grad(img_joint, cam_joint, joint_valid)
        self.optim.epoch = self.epoch
        self.optim.step(grad)
        return loss
