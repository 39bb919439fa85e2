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

    @classmethod
    def from_module(cls, module, faces, j_regressor, device='cuda', **kw):
        """Train the parameters of a gator_amd.models.GATOR module (reference state_dict layout)."""
        return cls({k: v.detach().cpu() for k, v in module.state_dict().items()}, M.consts_from_module(module, device), faces, j_regressor, device, **kw)

    def state_dict(self):
        """Reference-layout state_dict of the trained weights: module.load_state_dict(trainer.state_dict()) puts them on the
        inference kernels."""
        return self.params.state_dict()

    def loss_and_grad(self, pose2d, targets, training=True):
        P = self.params.views()
        mesh, pose3d = M.gator_forward(P, self.consts, pose2d, self.gen, self.rates, training, self.params.buffers)
        loss, parts = self.losses.total(mesh, pose3d, targets, with_edge=self.epoch > self.edge_loss_start)
        one = ops.raw_unary(ops.U_AFFINE, loss.detach(), 0.0, 1.0)          # d loss / d loss (autograd's default would be an aten fill)
        grad, = torch.autograd.grad(loss, self.params.flat, grad_outputs=one)
        return loss.detach(), {k: v.detach() for k, v in parts.items()}, grad

    def capture(self, pose2d, targets):
        """Capture forward + losses + backward of one step (this batch shape) into a hipGraph: the ~3 000 kernel launches of a step
        become one graph launch, the host only replays.  Dropout offsets and Adam's step index come from a device counter that the
        graph itself advances, so every replay draws new masks.  The gradient all-reduce and the Adam launch stay outside the graph."""
        dev = self.params.flat.device
        self.gen.device_steps(dev)
        self.optim.device_step = self.gen.counter
        self._x = pose2d.clone()
        self._tg = {k: v.clone() for k, v in targets.items()}
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                        # warm-up off the capture: workspaces and the allocator's pools
            for _ in range(2):
                self._body()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.gen.counter.zero_()
        self.optim.step_count = 0
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = self._body()
        return self

    def _body(self):
        self.gen.begin_step()
        return self.loss_and_grad(self._x, self._tg)

    def step(self, pose2d, targets):
        """optimizer.zero_grad(); loss.backward(); optimizer.step()  (base.py:151-153)"""
        if self._graph is not None:
            ops.raw_unary(ops.U_AFFINE, pose2d, 1.0, 0.0, out=self._x)
            for k, v in targets.items():
                ops.raw_unary(ops.U_AFFINE, v, 1.0, 0.0, out=self._tg[k])
            self._graph.replay()
            loss, parts, grad = self._out
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
