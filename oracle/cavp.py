"""Oracle: CAVP video encoder (SlowOnly-R50) forward, functional torch-fp32 restatement.  Test infrastructure only.

SURVEY.md section 8(f) row N1.  Operates on the reference-layout ``state_dict`` of ``CAVP_Inference`` (keys
``video_encoder.*`` / ``video_project_head.*``).  Follows

  * CAVP_Inference.encode_video           inference/model/cavp_model.py:47-65   (pool=False, normalize=True as called by
                                          Extract_CAVP_Features.forward, inference/demo_util.py:161)
  * ResNet3dSlowOnly defaults             inference/model/cavp_modules.py:1233-1268 (conv1 (1,7,7), inflate (0,0,1,1),
                                          no pool2, lateral=False)
  * ResNet3d stem / stages / forward      cavp_modules.py:757-779, 484-518, 837-859 (depth 50: Bottleneck3d x (3,4,6,3),
                                          spatial strides (1,2,2,2), temporal strides 1, AdaptiveAvgPool2d((1,1)) on the
                                          last stage)
  * Bottleneck3d                          cavp_modules.py:167-330 ('pytorch' style: the stride sits on conv2; inflate ->
                                          conv1 is (3,1,1) pad (1,0,0); conv2 (1,3,3) pad (0,1,1); conv3 1x1x1 without
                                          activation; out = relu(conv3 + identity/downsample))
  * downsample                            cavp_modules.py:1006-1016 (1x1x1 conv, stride (1,s,s), BN, no activation)

PARITY PIN: ``mmcv`` (1.7.1, third-party, not under /root/reference and not installed) provides ``ConvModule``; its
documented behaviour for these configs -- Conv3d(bias=False) -> BatchNorm3d (eval: running statistics, eps 1e-5) ->
ReLU when act_cfg is set -- is restated in ``_cm`` below.  tests/golden/make_golden.py --cavp imports the reference's
own ResNet3dSlowOnly with a declared stand-in for that one class to check the TOPOLOGY (kernel sizes, strides,
inflation, downsample placement, key names) of this restatement; numerically the pin is "this restatement", i.e.
parity is unpinned at the mmcv boundary."""
import torch
import torch.nn.functional as F

STAGE_BLOCKS = (3, 4, 6, 3)
STAGE_INFLATE = (0, 0, 1, 1)
STAGE_STRIDE = (1, 2, 2, 2)
BN_EPS = 1e-5


def _cm(sd, p, x, stride=(1, 1, 1), padding=(0, 0, 0), act=True):
    """mmcv ConvModule(conv3d bias=False, BN3d, ReLU?) in eval mode."""
    y = F.conv3d(x, sd[p + ".conv.weight"], None, stride=stride, padding=padding)
    y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                     sd[p + ".bn.bias"], False, 0.0, BN_EPS)
    return F.relu(y) if act else y


def bottleneck(sd, p, x, stride, inflate):
    idt = x
    if inflate:
        out = _cm(sd, p + ".conv1", x, padding=(1, 0, 0))                   # (3,1,1)
    else:
        out = _cm(sd, p + ".conv1", x)                                       # (1,1,1)
    out = _cm(sd, p + ".conv2", out, stride=(1, stride, stride), padding=(0, 1, 1))   # (1,3,3)
    out = _cm(sd, p + ".conv3", out, act=False)
    if (p + ".downsample.conv.weight") in sd:
        idt = _cm(sd, p + ".downsample", x, stride=(1, stride, stride), act=False)
    return F.relu(out + idt)


@torch.no_grad()
def backbone(sd, x, stage_blocks=STAGE_BLOCKS):
    """x (B,3,T,H,W) -> (B,C,T,1,1)."""
    x = _cm(sd, "video_encoder.conv1", x.float(), stride=(1, 2, 2), padding=(0, 3, 3))
    x = F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
    for li, nb in enumerate(stage_blocks):
        for bi in range(nb):
            x = bottleneck(sd, f"video_encoder.layer{li + 1}.{bi}", x, STAGE_STRIDE[li] if bi == 0 else 1,
                           STAGE_INFLATE[li])
    return x.mean(dim=(3, 4), keepdim=True)


@torch.no_grad()
def encode_video(sd, video, normalize=True, stage_blocks=STAGE_BLOCKS):
    """video (B,T,3,H,W) in [0,1] -> (B,T,embed_dim).  pool=False path of CAVP_Inference.encode_video."""
    f = backbone(sd, video.permute(0, 2, 1, 3, 4), stage_blocks)
    bs, c, t = f.shape[:3]
    f = f.reshape(bs, c, t).permute(0, 2, 1)
    f = F.linear(f, sd["video_project_head.weight"], sd["video_project_head.bias"])
    return F.normalize(f, dim=-1) if normalize else f
