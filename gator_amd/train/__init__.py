"""The training row: the reference's training step (lib/core/base.py:122-183) on HIP kernels.
ops: differentiable primitives; model: GATOR in training mode; losses / optim / trainer: the rest of the step."""
