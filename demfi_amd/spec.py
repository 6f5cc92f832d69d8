"""Layer table of the DeMFI-Net_rb forward path.

The product keeps the reference's ``state_dict`` contract (260 tensors):
every convolution of /root/reference/DeMFInet.py is listed here by its state_dict prefix with
(cout, cin, kh, kw).  Conv3d layers of D1 (kernel (1,3,3), DeMFInet.py:30-34, 524-542) are flagged
so their weight keeps the 5-D shape ``[cout, cin, 1, 3, 3]`` in the state_dict.

Nothing here touches the device; it is plain host-side metadata shared by the module surface,
the weight repacker and the synthetic-weight generator.
"""
from collections import OrderedDict


class HyperParams:
    """The constructor arguments the reference reads from ``args`` (DeMFInet.py:17-21, 32, 42, 326, 328)."""

    def __init__(self, gpu=0, nf=64, scale_factor=2, num_ResB_FACFB=5, num_ResB_Dec=5,
                 shared_FGAC_flag=True, visualization_flag=False, fgac_rr=0, fgac_sr=0, fgac_map=0):
        self.gpu = gpu
        self.nf = nf
        self.scale_factor = scale_factor
        self.num_ResB_FACFB = num_ResB_FACFB
        self.num_ResB_Dec = num_ResB_Dec
        self.shared_FGAC_flag = shared_FGAC_flag
        self.visualization_flag = visualization_flag
        # radii of the generalised FGAC (function-local constants 0 at DeMFInet.py:401-402) and its index map
        # (0: as the reference code computes it, 1: pixel-centred window) -- extensions, defaults = the released model
        self.fgac_rr, self.fgac_sr, self.fgac_map = fgac_rr, fgac_sr, fgac_map
        # engine-level switch: also compute the maps of the visualisation / training return tuples (DeMFInet.py:167-176, 454-496);
        # DeMFInet.forward turns it on for args.visualization_flag and for is_training calls
        self.extras = bool(visualization_flag)


def layer_table(hp=None):
    """OrderedDict prefix -> (cout, cin, kh, kw, is_conv3d) in the reference's registration order."""
    hp = hp or HyperParams()
    nf = hp.nf
    r2 = hp.scale_factor * hp.scale_factor
    G0, G, C, D = 96, 32, 4, 12           # FF_RDB defaults, DeMFInet.py:190-194
    t = OrderedDict()

    def add(name, cout, cin, kh, kw, c3d=False):
        t[name] = (cout, cin, kh, kw, c3d)

    p = 'FF_RDB_Module.'
    add(p + 'SFENet1', G0, 12 * r2, 5, 5)
    add(p + 'SFENet2', G0, G0, 3, 3)
    for i in range(D):
        for c in range(C):
            add(p + 'RDBs.%d.convs.%d.conv.0' % (i, c), G, G0 + c * G, 3, 3)
        add(p + 'RDBs.%d.LFF' % i, G0, G0 + C * G, 1, 1)
    add(p + 'GFF.0', G0, D * G0, 1, 1)
    add(p + 'GFF.1', G0, G0, 3, 3)
    add(p + 'UPNet.0', 256, G0, 3, 3)
    add(p + 'UPNet.2', 2 * nf + 5, 64, 3, 3)

    p = 'FAC_FB_Module.'
    add(p + 'conv_first', nf, nf, 3, 3)
    for i in range(hp.num_ResB_FACFB):
        add(p + 'feature_extraction.%d.conv1' % i, nf, nf, 3, 3)
        add(p + 'feature_extraction.%d.conv2' % i, nf, nf, 3, 3)
    fgacs = ['shared_FGAC'] if hp.shared_FGAC_flag else ['FGAC_F1toF0', 'FGAC_F0toF1']
    for f in fgacs:
        add(p + f + '.conv_ref_k', nf, nf, 1, 1)
        add(p + f + '.conv_source_k', nf, nf, 1, 1)
        add(p + f + '.w_gen', nf, 2 * nf, 3, 3)
        add(p + f + '.w_gen_2', 1, nf, 3, 3)
        add(p + f + '.fusion', nf, nf, 1, 1)

    p = 'Refine_Module.'
    add(p + 'enc1', nf, 3 * nf + 9, 4, 4)
    add(p + 'enc2', 2 * nf, nf, 4, 4)
    add(p + 'enc3', 4 * nf, 2 * nf, 4, 4)
    add(p + 'dec0', 4 * nf, 4 * nf, 3, 3)
    add(p + 'dec1', 2 * nf, 6 * nf, 3, 3)
    add(p + 'dec2', nf, 3 * nf, 3, 3)
    add(p + 'dec3', 2 * nf + 5, nf, 3, 3)

    add('Dec_first', nf, nf, 3, 3, True)
    for i in range(hp.num_ResB_Dec):
        add('Decoder_res.%d.conv1' % i, nf, nf, 3, 3, True)
        add('Decoder_res.%d.conv2' % i, nf, nf, 3, 3, True)
    add('Dec_last1', nf, nf, 3, 3, True)
    add('Dec_last2', 3, nf, 3, 3, True)

    add('Ch_Reducer', nf, 3 * nf, 7, 7)
    p = 'Booster_Module.'
    add(p + 'Mixer.conv_ref1', nf // 2, 30, 7, 7)
    add(p + 'Mixer.conv_ref2', nf // 2, nf // 2, 3, 3)
    add(p + 'Mixer.conv_delta1', nf // 2, 5, 7, 7)
    add(p + 'Mixer.conv_delta2', nf // 2, nf // 2, 3, 3)
    add(p + 'Mixer.conv_blend1', nf // 2, nf, 3, 3)
    add(p + 'Mixer.conv_blend2', nf, nf // 2, 3, 3)
    for g in ('z', 'r', 'q'):
        add(p + 'GB.conv%s1' % g, nf, 2 * nf, 1, 5)
    for g in ('z', 'r', 'q'):
        add(p + 'GB.conv%s2' % g, nf, 2 * nf, 5, 1)
    add(p + 'flow_occ.conv1', nf // 2, nf, 3, 3)
    add(p + 'flow_occ.conv2', 5, nf // 2, 3, 3)

    add('Dec_first_2', nf, 9 + nf + 9 + 5 + 12, 3, 3)
    for i in range(hp.num_ResB_Dec):
        add('Decoder_res_2.%d.conv1' % i, nf, nf, 3, 3)
        add('Decoder_res_2.%d.conv2' % i, nf, nf, 3, 3)
    add('Dec_last1_2', nf, nf, 3, 3)
    add('Dec_last2_2', 9, nf, 3, 3)
    return t


def weight_shape(entry):
    cout, cin, kh, kw, c3d = entry
    return (cout, cin, 1, kh, kw) if c3d else (cout, cin, kh, kw)


def state_dict_shapes(hp=None):
    """OrderedDict key -> shape, 260 entries for the default hyper-parameters."""
    out = OrderedDict()
    for name, e in layer_table(hp).items():
        out[name + '.weight'] = weight_shape(e)
        out[name + '.bias'] = (e[0],)
    return out
