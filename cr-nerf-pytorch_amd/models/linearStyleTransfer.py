"""Cross-ray appearance transfer + decoder with the reference's class names, constructor
signatures and state_dict keys (reference models/linearStyleTransfer.py: CNN :6-37, MulLayer :43-94,
style_net :278-291), computed by the HIP kernels of csrc/crossray.hip.

The modules only own parameters; style_net.forward drives the kernel sequence
chansum -> (mean) -> gram -> matrix -> fold -> apply.  When a torch.distributed process group is
passed (rays sharded across GPUs), the two tiny reductions become RCCL all-reduces (parallel.py)."""
import torch
from torch import nn

from .. import ops
from .nerf_decoder_stylenerf import NeuralRenderer


def _pixel_major(x):
    """[1,C,H,W] -> ([HW,C] contiguous, (H,W)).  A grid built the reference way
    (feature[R,64] -> transpose -> view, eval.py:291-292) is already pixel-major in memory, so this
    is a zero-copy view; a plain NCHW tensor is transposed once."""
    if x.dim() != 4 or x.shape[0] != 1:
        raise ValueError("expected a [1,C,H,W] grid, got %s" % (tuple(x.shape),))
    _, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(H * W, C).contiguous(), (H, W)


class CNN(nn.Module):
    def __init__(self, matrixSize=32, in_channel=64):
        super().__init__()
        if matrixSize != 32 or in_channel != 64:
            raise NotImplementedError("crnerf_amd: CNN is implemented for matrixSize=32, in_channel=64")
        self.convs = nn.Sequential(nn.Conv2d(in_channel, 128, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True),
                                   nn.Conv2d(128, 64, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True),
                                   nn.Conv2d(64, matrixSize, 1, 1, 0))
        self.fc = nn.Linear(matrixSize * matrixSize, matrixSize * matrixSize)

    def conv_tensors(self):
        c0, c2, c4 = self.convs[0], self.convs[2], self.convs[4]
        return [c0.weight.reshape(128, 64), c0.bias, c2.weight.reshape(64, 128), c2.bias, c4.weight.reshape(32, 64), c4.bias]

    def gram_sum(self, x_pm, mean):
        """Sum over pixels of f(x-mean) f(x-mean)^T (before the /(h*w), CNN.forward :31-34)."""
        return ops.crossray_gram(x_pm, mean, self.conv_tensors())

    def matrix(self, gram_sum, count):
        return ops.crossray_matrix(gram_sum, count, self.fc.weight, self.fc.bias)

    def forward(self, x):
        """x: centred [1,64,h,w] -> [1,1024] (reference CNN.forward)."""
        xp, (H, W) = _pixel_major(x)
        zero = torch.zeros(64, device=x.device)
        return self.matrix(self.gram_sum(xp, zero), H * W).view(1, -1)


class MulLayer(nn.Module):
    def __init__(self, matrixSize=32, in_channel=64):
        super().__init__()
        self.snet = CNN(matrixSize)
        self.cnet = CNN(matrixSize)
        self.matrixSize = matrixSize
        self.compress = nn.Conv2d(in_channel, matrixSize, 1, 1, 0)
        self.unzip = nn.Conv2d(matrixSize, in_channel, 1, 1, 0)
        self.transmatrix = None

    def lin_tensors(self):
        return [self.compress.weight.reshape(32, 64), self.compress.bias, self.unzip.weight.reshape(64, 32), self.unzip.bias]


class encoder_sameoutputsize(nn.Module):
    """Appearance encoder -- reference models/linearStyleTransfer.py:208-276 (same ctor, attribute names and
    state_dict keys conv1..conv7).  Inference and training both run the HIP kernels (csrc/encoder.hip,
    csrc/encoder_train.hip); the nn.Conv2d / pooling members only hold the parameters under the reference's names."""

    def __init__(self, out_channel=64):
        super().__init__()
        if out_channel != 64:
            raise NotImplementedError("crnerf_amd: encoder_sameoutputsize is implemented for out_channel=64")
        self.conv1 = nn.Conv2d(3, 3, 1, 1, 0)
        self.reflecPad1 = nn.ReflectionPad2d((1, 1, 1, 1))
        self.conv2 = nn.Conv2d(3, 64, 3, 1, 0)
        self.relu2 = nn.LeakyReLU(0.2, inplace=True)
        self.reflecPad3 = nn.ReflectionPad2d((1, 1, 1, 1))
        self.conv3 = nn.Conv2d(64, 64, 3, 1, 0)
        self.relu3 = nn.LeakyReLU(0.2, inplace=True)
        self.maxPool = nn.MaxPool2d(kernel_size=2, stride=2, return_indices=True)
        self.reflecPad4 = nn.ReflectionPad2d((1, 1, 1, 1))
        self.conv4 = nn.Conv2d(64, 128, 3, 1, 0)
        self.relu4 = nn.LeakyReLU(0.2, inplace=True)
        self.reflecPad5 = nn.ReflectionPad2d((1, 1, 1, 1))
        self.conv5 = nn.Conv2d(128, 128, 3, 1, 0)
        self.relu5 = nn.LeakyReLU(0.2, inplace=True)
        self.maxPool2 = nn.MaxPool2d(kernel_size=2, stride=2, return_indices=True)
        self.reflecPad6 = nn.ReflectionPad2d((1, 1, 1, 1))
        self.conv6 = nn.Conv2d(128, 128, 3, 1, 0)
        self.relu6 = nn.LeakyReLU(0.2, inplace=True)
        self.adppool = nn.AdaptiveAvgPool2d(32)
        self.conv7 = nn.Conv2d(128, out_channel, 1, 1, 0)
        self.relu7 = nn.LeakyReLU(0.2, inplace=True)

    def forward(self, x):
        if x.dim() != 4 or x.shape[0] != 1 or x.shape[1] != 3:
            raise ValueError("encoder_sameoutputsize expects [1,3,H,W], got %s" % (tuple(x.shape),))
        weights = ops.cached_list(self, "conv_tensors", lambda: [t for c in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6,
                                                                             self.conv7) for t in (c.weight, c.bias)])
        if torch.is_grad_enabled() and (x.requires_grad or ops.any_requires_grad(self)):
            from ..autograd import EncoderFn   # training: HIP forward-with-save + HIP backward (csrc/encoder_train.hip)
            grid = EncoderFn.apply(x.to(torch.float32).contiguous(), *weights)
            return grid.view(1, 32, 32, 64).permute(0, 3, 1, 2)
        grid = ops.encoder_forward(x, weights)       # [1024,64] pixel-major
        return grid.view(1, 32, 32, 64).permute(0, 3, 1, 2)                                   # NCHW view, zero-copy for style_net


class style_net(nn.Module):
    def __init__(self, args, residual_blocks=2):
        super().__init__()
        nerf_channel = args.nerf_out_dim
        self.multi_net = MulLayer(in_channel=nerf_channel)
        self.decoder = NeuralRenderer(img_size=(args.img_wh[0], args.img_wh[1]), featmap_size=(args.img_wh[0], args.img_wh[1]),
                                      feat_nc=args.nerf_out_dim, out_dim=3, args_here=args)

    def affine_from_stats(self, c_sum, c_gram, c_count, style_feature, kernels=None):
        """Everything downstream of the two global reductions: replicated, tiny.  c_sum[64] and
        c_gram[1024] are GLOBAL sums over all content pixels, c_count the global pixel count."""
        k = kernels or ops
        mn = self.multi_net
        sp, _ = _pixel_major(style_feature)
        s_mean = (k.crossray_chansum(sp) / sp.shape[0]).contiguous()
        s_gram = k.crossray_gram(sp, s_mean, mn.snet.conv_tensors())
        s_matrix = k.crossray_matrix(s_gram, sp.shape[0], mn.snet.fc.weight, mn.snet.fc.bias)
        c_mean = (c_sum / c_count).contiguous()
        c_matrix = k.crossray_matrix(c_gram, c_count, mn.cnet.fc.weight, mn.cnet.fc.bias)
        return k.crossray_fold(s_matrix, c_matrix, c_mean, s_mean, mn.lin_tensors() + list(self.decoder.rgb_tensors()))

    def forward(self, content_feature, style_feature, type=None):
        train = torch.is_grad_enabled() and (content_feature.requires_grad or ops.any_requires_grad(self))
        if style_feature is None and type == "content":
            if train:   # decoder only: sigmoid(1x1 conv), nerf_decoder_stylenerf.py:279-291 (n_blocks == 0)
                from ..autograd import ContentDecoderFn
                w, b = self.decoder.rgb_tensors()
                xp, (H, W) = _pixel_major(content_feature)
                return ContentDecoderFn.apply(xp, w, b, self.decoder_tensors()).view(1, 3, H, W)
            return self.decoder(content_feature)
        xp, (H, W) = _pixel_major(content_feature)
        sp, _ = _pixel_major(style_feature)
        if train:
            from ..autograd import DecoderFn   # HIP forward + HIP backward (crnerf_crossray_decode_backward_f32)
            return DecoderFn.apply(xp, sp, *self.decoder_tensors()).view(1, 3, H, W)
        return ops.crossray_decode(xp, sp, self.decoder_tensors()).view(1, 3, H, W)

    def decoder_tensors(self):
        """The 22 parameter tensors in state_dict order (crnerf_crossray_decode_f32's `weights`): the 1 x 1 convolutions' [cout, cin, 1, 1] weights as
        [cout, cin] views.  The Parameter objects are looked up once per module (ops.cached_list: a step asks four times, through ~140 attribute
        lookups each); the views are made per call -- every use of the decoder keeps its own view node, so gradients accumulate use by use in
        the same order with and without deferred accumulation (tests/test_gpu_train_aux.py: bit-equal)."""
        def build():
            mn = self.multi_net
            convs = lambda net: [t for i in (0, 2, 4) for t in (net.convs[i].weight, net.convs[i].bias)]  # noqa: E731
            rgb = self.decoder.feat_2_rgb_list[0]
            return (convs(mn.snet) + [mn.snet.fc.weight, mn.snet.fc.bias] + convs(mn.cnet) + [mn.cnet.fc.weight, mn.cnet.fc.bias]
                    + [mn.compress.weight, mn.compress.bias, mn.unzip.weight, mn.unzip.bias, rgb.weight, rgb.bias])
        return [p.reshape(p.shape[0], p.shape[1]) if p.dim() == 4 else p for p in ops.cached_list(self, "decoder_params", build)]
