"""Drop-in for architectures/losses.py (`calc_iou` :6-23 is fused into the loss kernel;
`FocalLoss` :25-168)."""
import torch.nn as nn

from .. import ops_detect


class FocalLoss(nn.Module):
    """forward(classifications [B,A,C] probs, regressions [B,A,4], anchors [1,A,4],
    annotations [B,M,5]) -> (cls_loss [1], reg_loss [1]) with grad — losses.py:30, 168.
    One fused HIP pass (IoU assignment + focal + smooth-L1) instead of a per-image Python loop
    of ~45 ATen launches with >= 8 [A,C] temporaries."""

    def forward(self, classifications, regressions, anchors, annotations):
        cl, rl, argmax, state, num_pos = ops_detect.FocalLossFn.apply(
            classifications, regressions, anchors, annotations)
        self.last_assignment = (argmax, state, num_pos)     # integer intermediates, for parity tests
        return cl, rl
