"""ResNets (18/34/50/101/152, v1.5 bottlenecks) and VGG-16 (plain and _bn) defined in-repo: torchvision is not available on the target image,
and the harness only needs the TOPOLOGY (random weights) to drive the quantization path with the
real sequence of layer outputs.

Layers are created through `nn.Conv2d` / `nn.Linear` / ... looked up at construction time, so
building a model inside `with QM(args, qparams):` yields the patched *WithId layers exactly like
torchvision models do in the reference (inference_sim.py:170)."""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)   # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class ResNet(nn.Module):
    """torchvision-style ResNet: `block` with `layers` blocks per stage."""

    def __init__(self, block=Bottleneck, layers=(3, 4, 6, 3), num_classes=1000):
        super().__init__()
        self.block = block
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, planes, blocks, stride=1):
        block, down = self.block, None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def ResNet50(num_classes=1000):
    return ResNet(Bottleneck, (3, 4, 6, 3), num_classes)


def mark_before_relu(model):
    """utils/mark_relu.py:4-29: the stem conv and conv1/conv2 (+ their BNs) of every bottleneck, conv1 (+ bn1)
    of every basic block feed a ReLU directly -> half-range quantization; the block's last conv and the
    downsample convs do not."""
    if isinstance(model, ResNet):
        model.conv1.before_relu = True
    for m in model.modules():
        if isinstance(m, Bottleneck):
            for sub in (m.conv1, m.bn1, m.conv2, m.bn2):
                sub.before_relu = True
        elif isinstance(m, BasicBlock):
            m.conv1.before_relu = True
            m.bn1.before_relu = True


VGG16_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')


class VGG16(nn.Module):
    """torchvision-style VGG-16 (no BN): 13 conv outputs; with arch 'vgg16' the manager treats the
    ReLUs as fused (force_positive, inference_quantization_manager.py:492)."""

    def __init__(self, num_classes=1000, fc_width=4096, batch_norm=False):
        super().__init__()
        layers, cin = [], 3
        for v in VGG16_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers.append(nn.Conv2d(cin, v, 3, padding=1))
                if batch_norm:
                    layers.append(nn.BatchNorm2d(v))
                layers.append(nn.ReLU(inplace=True))
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 49, fc_width), nn.ReLU(True), nn.Linear(fc_width, fc_width),
                                        nn.ReLU(True), nn.Linear(fc_width, num_classes))

    def forward(self, x):
        x = self.avgpool(self.features(x))
        return self.classifier(torch.flatten(x, 1))


MODELS = {
    'resnet18': lambda: ResNet(BasicBlock, (2, 2, 2, 2)), 'resnet34': lambda: ResNet(BasicBlock, (3, 4, 6, 3)),
    'resnet50': ResNet50, 'resnet101': lambda: ResNet(Bottleneck, (3, 4, 23, 3)),
    'resnet152': lambda: ResNet(Bottleneck, (3, 8, 36, 3)),
    'vgg16': VGG16, 'vgg16_bn': lambda: VGG16(batch_norm=True),
}
