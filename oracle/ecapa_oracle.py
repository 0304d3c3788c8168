"""CPU ORACLE — test infrastructure, NOT a product path.

PyTorch-CPU float32 eager restatement of mindaudio/models/ecapatdnn.py (EcapaTDNN forward, eval mode).  Only tests/,
__graft_entry__.smoke() and benchmark baselines may import it.

**Parity unpinned**: the reference model is pure mindspore.nn (not runnable here) and has no test or golden vector.
What pins this file is the source it restates, including the quirks:
  * TDNNBlock order is conv -> ReLU -> BatchNorm (ecapatdnn.py:60-64);
  * nn.Conv1d default pad_mode "same": zero padding, output length = input length (ecapatdnn.py:47-56);
  * MyBatchNorm1d ignores its eps / momentum arguments and wraps a default BatchNorm2d (eps 1e-5) (ecapatdnn.py:7-32);
  * the SE block averages over all T frames, `lengths` is unused (ecapatdnn.py:152-156);
  * attentive statistics pooling without global context: softmax over T, std = sqrt(clip(sum w (x - mu)^2, 1e-12))
    (ecapatdnn.py:284-303); Res2Net: y_0 = x_0, y_1 = f_0(x_1), y_i = f_{i-1}(x_i + y_{i-1}) (ecapatdnn.py:100-113).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class TDNNBlock(nn.Module):
    def __init__(self, cin, cout, kernel_size, dilation):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, kernel_size, dilation=dilation, padding=dilation * (kernel_size - 1) // 2)
        self.norm = nn.BatchNorm1d(cout, eps=1e-5)

    def forward(self, x):
        return self.norm(F.relu(self.conv(x)))


class Res2NetBlock(nn.Module):
    def __init__(self, channels, scale=8, kernel_size=3, dilation=1):
        super().__init__()
        self.scale = scale
        self.blocks = nn.ModuleList([TDNNBlock(channels // scale, channels // scale, kernel_size, dilation)
                                     for _ in range(scale - 1)])

    def forward(self, x):
        ys, y = [], None
        for i, xi in enumerate(torch.chunk(x, self.scale, dim=1)):
            if i == 0:
                y = xi
            elif i == 1:
                y = self.blocks[0](xi)
            else:
                y = self.blocks[i - 1](xi + y)
            ys.append(y)
        return torch.cat(ys, dim=1)


class SEBlock(nn.Module):
    def __init__(self, cin, se_channels, cout):
        super().__init__()
        self.conv1 = nn.Conv1d(cin, se_channels, 1)
        self.conv2 = nn.Conv1d(se_channels, cout, 1)

    def forward(self, x):
        s = x.mean(2, keepdim=True)
        return torch.sigmoid(self.conv2(F.relu(self.conv1(s)))) * x


class SERes2NetBlock(nn.Module):
    def __init__(self, cin, cout, res2net_scale=8, se_channels=128, kernel_size=1, dilation=1):
        super().__init__()
        assert cin == cout, "the shipped configurations never take the shortcut convolution (ecapatdnn.py:229-236)"
        self.tdnn1 = TDNNBlock(cin, cout, 1, 1)
        self.res2net_block = Res2NetBlock(cout, res2net_scale, kernel_size, dilation)
        self.tdnn2 = TDNNBlock(cout, cout, 1, 1)
        self.se_block = SEBlock(cout, se_channels, cout)

    def forward(self, x):
        return self.se_block(self.tdnn2(self.res2net_block(self.tdnn1(x)))) + x


class AttentiveStatisticsPooling(nn.Module):
    def __init__(self, channels, attention_channels=128):
        super().__init__()
        self.tdnn = TDNNBlock(channels, attention_channels, 1, 1)
        self.conv = nn.Conv1d(attention_channels, channels, 1)

    def forward(self, x):
        w = torch.softmax(self.conv(torch.tanh(self.tdnn(x))), dim=2)
        mean = (w * x).sum(2)
        std = torch.sqrt((w * (x - mean.unsqueeze(2)) ** 2).sum(2).clamp(min=1e-12))
        return torch.cat((mean, std), dim=1).unsqueeze(2)


class EcapaTDNN(nn.Module):
    def __init__(self, input_size, lin_neurons=192, channels=(512, 512, 512, 512, 1536), kernel_sizes=(5, 3, 3, 3, 1),
                 dilations=(1, 2, 3, 4, 1), attention_channels=128, res2net_scale=8, se_channels=128):
        super().__init__()
        self.blocks = nn.ModuleList([TDNNBlock(input_size, channels[0], kernel_sizes[0], dilations[0])])
        for i in range(1, len(channels) - 1):
            self.blocks.append(SERes2NetBlock(channels[i - 1], channels[i], res2net_scale, se_channels, kernel_sizes[i],
                                              dilations[i]))
        self.mfa = TDNNBlock(channels[-1], channels[-1], kernel_sizes[-1], dilations[-1])
        self.asp = AttentiveStatisticsPooling(channels[-1], attention_channels)
        self.asp_bn = nn.BatchNorm1d(channels[-1] * 2, eps=1e-5)
        self.fc = nn.Conv1d(channels[-1] * 2, lin_neurons, 1)

    def forward(self, x):
        """x (B, T, F) -> (B, lin_neurons) (ecapatdnn.py:411-432; squeeze() of the trailing length-1 axis)."""
        x = x.transpose(1, 2)
        xl = []
        for layer in self.blocks:
            x = layer(x)
            xl.append(x)
        x = self.mfa(torch.cat(xl[1:], dim=1))
        x = self.asp_bn(self.asp(x))
        return self.fc(x).squeeze(2)


def forward_flops_per_utt(t=300, f=80, c=512, scale=8, att=128, se=128, lin=192):
    """Algorithmic forward FLOPs (2 * MACs) per utterance — SURVEY §8 cfg 5 (2.88 GFLOP at C=512, 10.78 at C=1024)."""
    cc = c // scale
    per_block = c * c + (scale - 1) * 3 * cc * cc + c * c
    macs = t * (5 * f * c + 3 * per_block + 3 * c * 3 * c + 3 * c * att + att * 3 * c)
    macs += 3 * (c * se + se * c) + 6 * c * lin
    return 2 * macs
