"""MinkowskiEngine.modules.resnet_block.BasicBlock as published (conv3-bn-relu, conv3-bn, (+ downsample(x) | x), relu;
attribute names conv1 / norm1 / conv2 / norm2 / downsample are what the checkpoint keys carry) -- stand-in, see
oracle/shims/README.md."""
import torch.nn as nn

import MinkowskiEngine as ME


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                             dimension=dimension)
        self.norm1 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                             dimension=dimension)
        self.norm2 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu(out._like(out.F + residual.F))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, *a, **k):
        raise NotImplementedError("Bottleneck is not used on the InsMOS path (MinkUNet14 uses BasicBlock)")
