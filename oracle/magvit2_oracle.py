"""Plain-torch restatement of the MAGVIT2 inference stacks -- TEST INFRASTRUCTURE, not product code.

Only tests/ (and tools that generate fixtures in the build container) may import this module; the product path
(1xgpt_amd/magvit2.py: HipDecoder / HipEncoder) never does.  It is functional: it runs the formulas below over the
PARAMETERS of any module tree that has the reference's attribute names (1xgpt_amd.magvit2.Decoder / Encoder are such
parameter holders), on whatever device / dtype those parameters live.

Pinned against outputs of the reference itself (tests/golden/magvit_small.npz, magvit_mid.npz, magvit_full.npz, made by
tools/make_goldens.py from /root/reference): tests/test_oracle_golden.py::test_magvit_oracle_*.

Formulas ("Tokens -> pixels"):
  swish(x) = x * sigmoid(x)                                                  improved_model.py:7-9
  ResBlock(x) = conv2(swish(GN32(conv1(swish(GN32(x)))))) + shortcut(x)      improved_model.py:12-51   (GN eps 1e-6)
  Upsampler(x) = depth_to_space_2(conv3x3(C -> 4C, bias)(x)), DCR order      improved_model.py:185-237
  Decoder / Encoder layer order                                              improved_model.py:124-182 / 54-121
"""
import torch
import torch.nn.functional as F


def swish(x):
    return x * torch.sigmoid(x)


def _gn(x, norm):
    return F.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps)


def _conv(x, conv):
    return F.conv2d(x, conv.weight, conv.bias, stride=conv.stride, padding=conv.padding)


def res_block(rb, x):
    h = _conv(swish(_gn(x, rb.norm1)), rb.conv1)
    h = _conv(swish(_gn(h, rb.norm2)), rb.conv2)
    if rb.in_filters != rb.out_filters:
        x = _conv(x, rb.nin_shortcut)
    return h + x


def depth_to_space(x, block_size):
    """DCR: channel (i*bs + j)*C' + c -> pixel (bs*h + i, bs*w + j), channel c."""
    c, h, w = x.shape[-3:]
    s = block_size ** 2
    assert c % s == 0
    outer = x.shape[:-3]
    x = x.reshape(-1, block_size, block_size, c // s, h, w).permute(0, 3, 4, 1, 5, 2)
    return x.reshape(*outer, c // s, h * block_size, w * block_size)


def decoder_forward(dec, z):
    z = _conv(z, dec.conv_in)
    for blk in dec.mid_block:
        z = res_block(blk, z)
    for i_level in reversed(range(dec.num_blocks)):
        for blk in dec.up[i_level].block:
            z = res_block(blk, z)
        if i_level > 0:
            z = depth_to_space(_conv(z, dec.up[i_level].upsample.conv1), 2)
    return _conv(swish(_gn(z, dec.norm_out)), dec.conv_out)


def encoder_forward(enc, x):
    x = _conv(x, enc.conv_in)
    for i_level in range(enc.num_blocks):
        for blk in enc.down[i_level].block:
            x = res_block(blk, x)
        if i_level < enc.num_blocks - 1:
            x = _conv(x, enc.down[i_level].downsample)
    for blk in enc.mid_block:
        x = res_block(blk, x)
    return _conv(swish(_gn(x, enc.norm_out)), enc.conv_out)


def bits_from_tokens(ids, codebook_dim=18):
    """(n, h, w) int64 -> (n, bits, h, w) in {-1, +1}; channel c = bit c (LSB first)."""
    sh = torch.arange(codebook_dim, device=ids.device).view(1, -1, 1, 1)
    return ((ids[:, None] >> sh) & 1).float() * 2 - 1


def rescale_u8(x):
    """visualize.py:84-92: (x + 1) * 127.5 in the tensor's own dtype, clamp [0, 255], truncate."""
    return torch.clamp((x + 1) * 127.5, 0, 255).to(torch.uint8)
