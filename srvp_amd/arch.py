"""
Layer tables of the two 64x64 encoder / decoder families (reference module/conv.py:157-224, 278-355), written as
data: one dict per conv block, consumed by the module builder (model.py: parameter containers with the reference's
state-dict keys) and by the HIP engine (engine.py: launch plans).

Block fields
  key / bnkey : state-dict prefixes of the conv weight and of its BatchNorm2d (None: block without BN)
  kind        : 'conv' (nn.Conv2d, OIHW weight) | 'convT' (nn.ConvTranspose2d, IOHW weight)
  cin, cout, k, s, p, act
  pre         : 'pool' if a MaxPool2d(2,2) precedes the block (conv.py:204,209,215,222)
  skip_out    : the block output is the stage output collected as skip connection (conv.py:148-150), index = stage
  cat         : index of the skip connection concatenated in front of the block (conv.py:270) or None
  post_up     : nearest x2 upsample after the block (conv.py:331,338,344,349)
  path        : module path elements used to rebuild the reference hierarchy
"""


def encoder_blocks(archi, nc, nh, nf):
    blocks = []
    if archi == 'dcgan':
        ch = [nc, nf, nf * 2, nf * 4, nf * 8]
        for i in range(4):
            blocks.append(dict(kind='conv', key=f'encoder.conv.{i}.0', bnkey=f'encoder.conv.{i}.1' if i else None,
                               cin=ch[i], cout=ch[i + 1], k=4, s=2, p=1, act='leaky_relu', pre=None, skip_out=i))
        blocks.append(dict(kind='conv', key='encoder.last_conv.0', bnkey='encoder.last_conv.1', cin=nf * 8, cout=nh,
                           k=4, s=1, p=0, act='tanh', pre=None, skip_out=None))
    elif archi == 'vgg':
        widths = [(nf, 2), (nf * 2, 2), (nf * 4, 3), (nf * 8, 3)]
        cin = nc
        for si, (wdt, n) in enumerate(widths):
            for li in range(n):
                idx = li if si == 0 else li + 1      # module 0 of stages 1..3 is the MaxPool2d
                blocks.append(dict(kind='conv', key=f'encoder.conv.{si}.{idx}.0', bnkey=f'encoder.conv.{si}.{idx}.1',
                                   cin=cin, cout=wdt, k=3, s=1, p=1, act='leaky_relu',
                                   pre='pool' if (si > 0 and li == 0) else None,
                                   skip_out=si if li == n - 1 else None))
                cin = wdt
        blocks.append(dict(kind='conv', key='encoder.last_conv.1.0', bnkey='encoder.last_conv.1.1', cin=nf * 8, cout=nh,
                           k=4, s=1, p=0, act='tanh', pre='pool', skip_out=None))
    else:
        raise ValueError(f"No encoder named '{archi}'")
    return blocks


def decoder_blocks(archi, nc, ny, nf, skip):
    c = 2 if skip else 1
    blocks = []
    if archi == 'dcgan':
        blocks.append(dict(kind='convT', key='decoder.first_upconv.0', bnkey='decoder.first_upconv.1', cin=ny,
                           cout=nf * 8, k=4, s=1, p=0, act='leaky_relu', cat=None, post_up=False))
        ch = [nf * 8, nf * 4, nf * 2, nf]
        for i in range(3):
            blocks.append(dict(kind='convT', key=f'decoder.conv.{i}.0', bnkey=f'decoder.conv.{i}.1', cin=ch[i] * c,
                               cout=ch[i + 1], k=4, s=2, p=1, act='leaky_relu', cat=i if skip else None, post_up=False))
        blocks.append(dict(kind='convT', key='decoder.conv.3', bnkey=None, cin=nf * c, cout=nc, k=4, s=2, p=1,
                           act='none', cat=3 if skip else None, post_up=False))
    elif archi == 'vgg':
        blocks.append(dict(kind='convT', key='decoder.first_upconv.0.0', bnkey='decoder.first_upconv.0.1', cin=ny,
                           cout=nf * 8, k=4, s=1, p=0, act='leaky_relu', cat=None, post_up=True))
        stages = [[nf * 8, nf * 8, nf * 4], [nf * 4, nf * 4, nf * 2], [nf * 2, nf], [nf]]
        cin = nf * 8
        for si, outs in enumerate(stages):
            for li, co in enumerate(outs):
                first = li == 0
                blocks.append(dict(kind='conv', key=f'decoder.conv.{si}.{li}.0', bnkey=f'decoder.conv.{si}.{li}.1',
                                   cin=cin * (c if first else 1), cout=co, k=3, s=1, p=1, act='leaky_relu',
                                   cat=si if (skip and first) else None,
                                   post_up=(si < 3 and li == len(outs) - 1)))
                cin = co
        blocks.append(dict(kind='convT', key='decoder.conv.3.1', bnkey=None, cin=nf, cout=nc, k=3, s=1, p=1, act='none',
                           cat=None, post_up=False))
    else:
        raise ValueError(f"No decoder named '{archi}'")
    return blocks
