"""Oracle (test infrastructure, NOT product): RFBNet-VGG forward on torch-CPU fp32.

A functional restatement of models/RFB_Net_vgg.py driven purely by a ``state_dict``
(the keys are the frozen contract) -- no nn.Module tree.  Stock torch CPU
ops in the reference's order:

  vgg base            models/RFB_Net_vgg.py:323-343, run :219-227
  BasicConv           :7-22      conv(no bias) -> BN(eps 1e-5) -> ReLU?
  BasicRFB_a (Norm)   :68-112
  BasicRFB  (extras)  :26-64, add_extras :354-378
  multibox heads      :387-416, run :238-248
  context pooling     :235-236, :242-244
  Context-Transformer :253-271
  eval softmaxes      :279-285

Size 512 + Context-Transformer has NO reference semantics (IndexError at :243, the
pooling lists have 6 entries for 7 sources).  The build defines the 7-entry list
``CTX_POOL[512]`` below; results on it are "parity unpinned (reference crashes)".
"""
import torch
import torch.nn.functional as F

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'C', 512, 512, 512, 'M', 512, 512, 512]
MBOX = {300: [6, 6, 6, 6, 4, 4], 512: [6, 6, 6, 6, 6, 4, 4]}
CTX_POOL = {300: [3, 2, 2, 2, 1, 1],            # models/RFB_Net_vgg.py:235-236
            512: [3, 2, 2, 2, 2, 1, 1]}         # build-defined (reference crashes)
BN_EPS = 1e-5


def _basic_conv(sd, pfx, x, stride=1, padding=0, dilation=1, relu=True, training=False):
    x = F.conv2d(x, sd[pfx + '.conv.weight'], None, stride, padding, dilation)
    if training:
        x = F.batch_norm(x, None, None, sd[pfx + '.bn.weight'], sd[pfx + '.bn.bias'], True, 0.01, BN_EPS)
    else:
        x = F.batch_norm(x, sd[pfx + '.bn.running_mean'], sd[pfx + '.bn.running_var'],
                         sd[pfx + '.bn.weight'], sd[pfx + '.bn.bias'], False, 0.01, BN_EPS)
    return F.relu(x) if relu else x


def _rfb_a(sd, pfx, x, scale=1.0, tr=False):
    """BasicRFB_a (models/RFB_Net_vgg.py:68-112), stride 1."""
    bc = lambda name, t, **kw: _basic_conv(sd, pfx + '.' + name, t, training=tr, **kw)
    x0 = bc('branch0.0', x)
    x0 = bc('branch0.1', x0, padding=1, relu=False)
    x1 = bc('branch1.0', x)
    x1 = bc('branch1.1', x1, padding=(1, 0))
    x1 = bc('branch1.2', x1, padding=3, dilation=3, relu=False)
    x2 = bc('branch2.0', x)
    x2 = bc('branch2.1', x2, padding=(0, 1))
    x2 = bc('branch2.2', x2, padding=3, dilation=3, relu=False)
    x3 = bc('branch3.0', x)
    x3 = bc('branch3.1', x3, padding=(0, 1))
    x3 = bc('branch3.2', x3, padding=(1, 0))
    x3 = bc('branch3.3', x3, padding=5, dilation=5, relu=False)
    out = bc('ConvLinear', torch.cat((x0, x1, x2, x3), 1), relu=False)
    short = bc('shortcut', x, relu=False)
    return F.relu(out * scale + short)


def _rfb(sd, pfx, x, stride, visual, scale=1.0, tr=False):
    """BasicRFB (models/RFB_Net_vgg.py:26-64)."""
    bc = lambda name, t, **kw: _basic_conv(sd, pfx + '.' + name, t, training=tr, **kw)
    v = visual
    x0 = bc('branch0.0', x, stride=stride)
    x0 = bc('branch0.1', x0, padding=v, dilation=v, relu=False)
    x1 = bc('branch1.0', x)
    x1 = bc('branch1.1', x1, stride=stride, padding=1)
    x1 = bc('branch1.2', x1, padding=v + 1, dilation=v + 1, relu=False)
    x2 = bc('branch2.0', x)
    x2 = bc('branch2.1', x2, padding=1)
    x2 = bc('branch2.2', x2, stride=stride, padding=1)
    x2 = bc('branch2.3', x2, padding=2 * v + 1, dilation=2 * v + 1, relu=False)
    out = bc('ConvLinear', torch.cat((x0, x1, x2), 1), relu=False)
    short = bc('shortcut', x, stride=stride, relu=False)
    return F.relu(out * scale + short)


def extras_plan(size):
    """add_extras (:354-378): list of ('rfb', stride, visual) / ('conv', k, pad)."""
    if size == 300:
        return [('rfb', 1, 2), ('rfb', 2, 2), ('rfb', 2, 2),
                ('conv', 1, 0), ('conv', 3, 0), ('conv', 1, 0), ('conv', 3, 0)]
    return [('rfb', 1, 2), ('rfb', 2, 2), ('rfb', 2, 2), ('rfb', 2, 1), ('rfb', 2, 1),
            ('conv', 1, 0), ('conv', 4, 1)]


def backbone(sd, x, size, training=False):
    """-> list of source feature maps (:219-233)."""
    idx = 0
    sources = []
    for v in VGG_CFG:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        elif v == 'C':
            x = F.max_pool2d(x, 2, 2, ceil_mode=True)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd['base.%d.weight' % idx], sd['base.%d.bias' % idx], 1, 1))
            idx += 2
        if idx == 23:                                   # after conv4_3 + ReLU
            sources.append(_rfb_a(sd, 'Norm', x, 1.0, training))
    assert idx == 30
    x = F.max_pool2d(x, 3, 1, 1)                        # pool5 = base.30
    x = F.relu(F.conv2d(x, sd['base.31.weight'], sd['base.31.bias'], 1, 6, 6))   # conv6
    x = F.relu(F.conv2d(x, sd['base.33.weight'], sd['base.33.bias']))            # conv7
    indicator = 3 if size == 300 else 5
    for k, item in enumerate(extras_plan(size)):
        if item[0] == 'rfb':
            x = _rfb(sd, 'extras.%d' % k, x, item[1], item[2], 1.0, training)
        else:
            x = _basic_conv(sd, 'extras.%d' % k, x, 1, item[2], 1, True, training)
        if k < indicator or k % 2 == 0:
            sources.append(x)
    return sources


def forward(sd, x, size, num_classes, phase=1, method='ours', setting='transfer',
            training=False, init=False, raw=False):
    """RFBNet.forward (:190-286).  num_classes = #foreground classes of the conf head.
    raw=True: eval-mode BatchNorm but WITHOUT the final softmaxes (the tensors the reference
    feeds to softmax at :282-284), for layer-level parity checks."""
    num = x.shape[0]
    sources = backbone(sd, x, size, training)
    ctx = (method == 'ours' and phase == 2)
    loc, conf, obj, pool = [], [], [], []
    for i, s in enumerate(sources):
        l = F.conv2d(s, sd['loc.%d.weight' % i], sd['loc.%d.bias' % i], 1, 1)
        c = F.conv2d(s, sd['conf.%d.weight' % i], sd['conf.%d.bias' % i], 1, 1)
        o = F.conv2d(s, sd['obj.%d.weight' % i], sd['obj.%d.bias' % i], 1, 1)
        loc.append(l.permute(0, 2, 3, 1).reshape(num, -1))
        conf.append(c.permute(0, 2, 3, 1).reshape(num, -1))
        obj.append(o.permute(0, 2, 3, 1).reshape(num, -1))
        if ctx:
            k = CTX_POOL[size][i]
            pool.append(F.max_pool2d(c, k, k, ceil_mode=True).permute(0, 2, 3, 1).reshape(num, -1))
    loc = torch.cat(loc, 1)
    conf = torch.cat(conf, 1)
    obj = torch.cat(obj, 1)
    if init:
        return conf.view(num, -1, num_classes)
    if ctx:
        conf = conf.view(num, -1, num_classes)
        cp = torch.cat(pool, 1).view(num, -1, num_classes)
        lin = lambda n, t: F.linear(t, sd[n + '.weight'], sd[n + '.bias'])
        if setting == 'incre':
            conf_base = lin('fc_base', conf) + conf
        theta = lin('theta', conf) + conf
        phi = lin('phi', cp) + cp
        g = lin('g', cp) + cp
        w = F.softmax(torch.matmul(theta, phi.transpose(1, 2)), dim=2)
        delta = torch.matmul(w, g) * sd['Wz']
        nov = conf + delta
        nov = nov / nov.norm(dim=2, keepdim=True)
        nov = F.linear(nov, sd['OBJ_Target.weight']) * sd['scale']
        conf = nov if setting == 'transfer' else torch.cat((conf_base, nov), dim=2)
    else:
        conf = conf.view(num, -1, num_classes)
    loc = loc.view(num, -1, 4)
    obj = obj.view(num, -1, 2)
    if training or raw:
        return loc, conf, obj
    return loc, F.softmax(conf, dim=-1), F.softmax(obj, dim=-1)


def context_block(sd, conf, cp, setting='transfer'):
    """Only the Context-Transformer block (:253-271) on given conf [B,P,C], pooled [B,M,C]."""
    lin = lambda n, t: F.linear(t, sd[n + '.weight'], sd[n + '.bias'])
    theta = lin('theta', conf) + conf
    phi = lin('phi', cp) + cp
    g = lin('g', cp) + cp
    w = F.softmax(torch.matmul(theta, phi.transpose(1, 2)), dim=2)
    nov = conf + torch.matmul(w, g) * sd['Wz']
    nov = nov / nov.norm(dim=2, keepdim=True)
    nov = F.linear(nov, sd['OBJ_Target.weight']) * sd['scale']
    if setting == 'incre':
        return torch.cat((lin('fc_base', conf) + conf, nov), dim=2)
    return nov


# ---------------------------------------------------------------------------
# deterministic, name-seeded synthetic weights -- shapes only, no reference code
# ---------------------------------------------------------------------------
def param_shapes(size, num_classes, phase=1, method='ours', setting='transfer'):
    """Ordered {key: shape} of the reference state_dict for (size, C, phase, setting)."""
    shapes = {}

    def bconv(pfx, cin, cout, k):
        kh, kw = (k, k) if isinstance(k, int) else k
        shapes[pfx + '.conv.weight'] = (cout, cin, kh, kw)
        shapes[pfx + '.bn.weight'] = (cout,)
        shapes[pfx + '.bn.bias'] = (cout,)
        shapes[pfx + '.bn.running_mean'] = (cout,)
        shapes[pfx + '.bn.running_var'] = (cout,)
        shapes[pfx + '.bn.num_batches_tracked'] = ()

    idx, cin = 0, 3
    for v in VGG_CFG:
        if v in ('M', 'C'):
            idx += 1
        else:
            shapes['base.%d.weight' % idx] = (v, cin, 3, 3)
            shapes['base.%d.bias' % idx] = (v,)
            cin = v
            idx += 2
    shapes['base.31.weight'] = (1024, 512, 3, 3)
    shapes['base.31.bias'] = (1024,)
    shapes['base.33.weight'] = (1024, 1024, 1, 1)
    shapes['base.33.bias'] = (1024,)
    # Norm = BasicRFB_a(512, 512)
    ip = 128
    bconv('Norm.branch0.0', 512, ip, 1); bconv('Norm.branch0.1', ip, ip, 3)
    bconv('Norm.branch1.0', 512, ip, 1); bconv('Norm.branch1.1', ip, ip, (3, 1)); bconv('Norm.branch1.2', ip, ip, 3)
    bconv('Norm.branch2.0', 512, ip, 1); bconv('Norm.branch2.1', ip, ip, (1, 3)); bconv('Norm.branch2.2', ip, ip, 3)
    bconv('Norm.branch3.0', 512, ip // 2, 1); bconv('Norm.branch3.1', ip // 2, (ip // 4) * 3, (1, 3))
    bconv('Norm.branch3.2', (ip // 4) * 3, ip, (3, 1)); bconv('Norm.branch3.3', ip, ip, 3)
    bconv('Norm.ConvLinear', 4 * ip, 512, 1); bconv('Norm.shortcut', 512, 512, 1)
    chans = [1024, 512, 256] if size == 300 else [1024, 512, 256, 256, 256]
    cin = 1024
    src_ch = [512]
    for k, cout in enumerate(chans):
        p = 'extras.%d' % k
        ip = cin // 8
        bconv(p + '.branch0.0', cin, 2 * ip, 1); bconv(p + '.branch0.1', 2 * ip, 2 * ip, 3)
        bconv(p + '.branch1.0', cin, ip, 1); bconv(p + '.branch1.1', ip, 2 * ip, 3); bconv(p + '.branch1.2', 2 * ip, 2 * ip, 3)
        bconv(p + '.branch2.0', cin, ip, 1); bconv(p + '.branch2.1', ip, (ip // 2) * 3, 3)
        bconv(p + '.branch2.2', (ip // 2) * 3, 2 * ip, 3); bconv(p + '.branch2.3', 2 * ip, 2 * ip, 3)
        bconv(p + '.ConvLinear', 6 * ip, cout, 1); bconv(p + '.shortcut', cin, cout, 1)
        cin = cout
        src_ch.append(cout)
    n = len(chans)
    if size == 300:
        bconv('extras.%d' % n, 256, 128, 1); bconv('extras.%d' % (n + 1), 128, 256, 3)
        bconv('extras.%d' % (n + 2), 256, 128, 1); bconv('extras.%d' % (n + 3), 128, 256, 3)
        src_ch += [256, 256]
    else:
        bconv('extras.%d' % n, 256, 128, 1); bconv('extras.%d' % (n + 1), 128, 256, 4)
        src_ch += [256]
    for name, mult in (('loc', 4), ('conf', num_classes), ('obj', 2)):
        for i, (c, m) in enumerate(zip(src_ch, MBOX[size])):
            shapes['%s.%d.weight' % (name, i)] = (m * mult, c, 3, 3)
            shapes['%s.%d.bias' % (name, i)] = (m * mult,)
    if method == 'ours' and phase == 2:
        d, t = (60, 20) if setting == 'transfer' else (15, 5)
        shapes['Wz'] = (d,)
        shapes['scale'] = (1,)
        if setting == 'incre':
            shapes['fc_base.weight'] = (d, d); shapes['fc_base.bias'] = (d,)
        for nme in ('theta', 'phi', 'g'):
            shapes[nme + '.weight'] = (d, d); shapes[nme + '.bias'] = (d,)
        shapes['OBJ_Target.weight'] = (t, d)
    return shapes
