"""Policy/value network of the reference (/root/reference/network.py:89-192), re-expressed for ROCm inference.

Same architecture, the same 252 state_dict keys and the same parameter-construction order (so
`torch.manual_seed(s); policyNN({})` reproduces the reference's initial weights and its checkpoints load):
  stem    conv1 3x3 (in_channels->256, no bias) + norm_layer (BN) + ReLU
  tower   resnet_blocks.{0..18}: conv1-bn1-ReLU-conv2-bn2 (+skip) -ReLU
  policy  conv_p1 1x1 -> p_norm1 -> ReLU -> conv_p2 1x1 (256->73, bias) -> flatten [73*8*8 = 4672]
  value   conv_v1 1x1 (256->1) -> v_norm -> ReLU -> flatten(64) -> fc_v1(64,256) -> ReLU -> fc_v2(256,1) -> tanh
forward(x, inference=False): softmax over the 4672 logits iff inference (network.py:190).

For self-play use `.to(memory_format=torch.channels_last)` + bf16 weights: the 3x3 convolutions are
implicit GEMMs with K = 2304 that MIOpen maps onto CDNA4 MFMA.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

WIDTH = 256
N_BLOCKS = 19
POLICY_PLANES = 73


class ResidualBlock(nn.Module):
    """BasicBlock of network.py:36-83 (stride 1, no downsample): attribute names fix the state_dict keys."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + x)


class policyNN(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        config = config or {}
        in_channels = config.get("in_channels", 119)
        # construction order below == RNG consumption order of the reference (network.py:105-137)
        self.conv1 = nn.Conv2d(in_channels, WIDTH, kernel_size=3, padding=1, bias=False)
        self.norm_layer = nn.BatchNorm2d(WIDTH)
        self.conv_p1 = nn.Conv2d(WIDTH, WIDTH, kernel_size=1, bias=False)
        self.p_norm1 = nn.BatchNorm2d(WIDTH)
        self.conv_p2 = nn.Conv2d(WIDTH, POLICY_PLANES, kernel_size=1)
        self.conv_v1 = nn.Conv2d(WIDTH, 1, kernel_size=1, bias=False)
        self.v_norm = nn.BatchNorm2d(1)
        self.fc_v1 = nn.Linear(64, 256)
        self.fc_v2 = nn.Linear(256, 1)
        self.resnet_blocks = nn.Sequential(*[ResidualBlock(WIDTH) for _ in range(N_BLOCKS)])

    def policy_head(self, x):
        x = F.relu(self.p_norm1(self.conv_p1(x)))
        return torch.flatten(self.conv_p2(x), start_dim=1)

    def value_head(self, x):
        x = F.relu(self.v_norm(self.conv_v1(x)))
        x = F.relu(self.fc_v1(torch.flatten(x, start_dim=1)))
        return torch.tanh(self.fc_v2(x))

    def forward(self, x, inference=False):
        x = F.relu(self.norm_layer(self.conv1(x)))
        x = self.resnet_blocks(x)
        policy = self.policy_head(x)
        value = self.value_head(x)
        if inference:
            policy = torch.softmax(policy, dim=1)
        return policy, value


FLOPS_PER_BOARD = 2_914_845_184     # forward multiply-adds x2, SURVEY.md §2 row 5 [probed]
