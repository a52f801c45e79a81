"""MI355X-native counterpart of dgcnn/model.py:9-106 -- `build(point_cloud, flags) -> logits`.

Same control flow as the reference (EdgeConv stack -> concat -> MergedEdgeConv -> global max ->
tile -> concat-all -> fc -> dropout -> Final).  Two things are laid out differently in HBM, with
identical results up to fp32 rounding:
  * tf.concat never copies: every block writes its outputs straight into column slices of one
    (B*N, sum(2F_i+64)+1024) buffer (channel order of model.py:83-85 minus the leading global block);
  * the tiled global feature (model.py:80-85: 1024 identical channels for all N points of a cloud)
    is not materialised: FC0 = x_local W[1024:] + (g_cloud W[:1024]) with the second term computed
    once per cloud and added in the GEMM epilogue (distributivity of the 1x1 convolution).
"""
from __future__ import annotations

from . import _engine as E
from . import ops


def build(point_cloud, flags):
    num_edge_conv = int(flags.EDGE_CONV_LAYERS)
    num_edge_filters = flags.EDGE_CONV_FILTERS
    num_fc = int(flags.FC_LAYERS)
    num_fc_filters = flags.FC_FILTERS
    is_training = bool(flags.TRAIN)
    k = int(flags.KVALUE)
    debug = bool(flags.DEBUG)
    num_class = int(flags.NUM_CLASS)
    c = E.ctx()

    if flags.MODEL_NAME not in ("dgcnn", "residual-dgcnn", "residual-dgcnn-nofc"):
        print("Unsupported MODEL_NAME: %s" % flags.MODEL_NAME)
        raise NotImplementedError("Unsupported MODEL_NAME: %s" % flags.MODEL_NAME)     # model.py:41-43

    x, B, N = E.as2d(point_cloud)
    R = B * N
    c.configure_slots(R)                                           # (DETERMINISTIC: one writer per statistics slot)
    ecf = ops._listify(num_edge_filters, num_edge_conv, "num_filters")
    nofc = flags.MODEL_NAME == "residual-dgcnn-nofc"

    plan = None
    if not nofc:
        # one buffer for [L0 max|mean|net, L1 ..., merged(1024)]  (model.py:83-85 without the tiled global)
        widths = []
        for f in ecf:
            widths += [2 * f, 64]
        ctot = sum(widths) + 1024
        if num_fc < 1:
            # FC_LAYERS = 0 (model.py:88 + ops.py:151: a loop of zero trips): dropout and Final act on the concat itself, so the tiled
            # global feature is materialised here -- `big` is the trailing ctot columns of the (R, 1024 + ctot) concat
            bigfull = c.new_buffer(R, 1024 + ctot)
            big = bigfull[:, 1024:]
        else:
            big = c.new_buffer(R, ctot)
        merged_in = c.new_buffer(R, 64 * num_edge_conv)          # model.py:60-63 concat of the conv1 outputs
        offs = []
        o = 0
        for f in ecf:
            offs.append(o)
            o += 2 * f + 64

        def plan(i):
            o = offs[i]
            f = ecf[i]
            return (big[:, o:o + 2 * f], big[:, o + 2 * f:o + 2 * f + 64]), merged_in[:, 64 * i:64 * (i + 1)]

    # parameter-only work of the whole step (conv0's folded weights; the head's weight planes) goes first, on the side stream
    pl_head = False
    if not nofc:
        fcf = ops._listify(num_fc_filters, num_fc, "num_filters")
        pl_head = (num_fc >= 1 and E.planes_ok(R, 64 * num_edge_conv, 1024) and E.planes_ok(R, ctot, fcf[0]) and ctot % 32 == 0 and
                   (num_fc < 2 or E.planes_ok(R, fcf[0], fcf[1])))
    if c.flat_param is not None and R >= E.SIDE_STREAM_MIN_ROWS:
        edge_w0, cin = [], x.shape[1]
        for i, f in enumerate(ecf):
            with E.variable_scope("EdgeConv%d" % i), E.variable_scope("conv0"):
                edge_w0.append((c.get_variable("weights", (2 * cin, f))[1], cin, f))
            cin = 64
        head_w = []
        if pl_head:
            kmax = max(ops._listify(k, num_edge_conv, "k"))
            c.ensure_plane_scales(R * kmax)
            with E.variable_scope("MergedEdgeConv"):
                head_w.append(c.get_variable("weights", (64 * num_edge_conv, 1024))[1])
            with E.variable_scope("FC0"):
                head_w.append(c.get_variable("weights", (1024 + ctot, fcf[0]))[1][1024:1024 + ctot])
            if num_fc >= 2:
                with E.variable_scope("FC1"):
                    head_w.append(c.get_variable("weights", (fcf[0], fcf[1]))[1])
        E.prepare_step_weights(edge_w0, head_w)

    if flags.MODEL_NAME == "dgcnn":
        tensors = ops.repeat_edge_conv(point_cloud, repeat=num_edge_conv, k=k, num_filters=num_edge_filters,
                                       trainable=is_training, debug=debug, _plan=plan)
    else:
        tensors = ops.repeat_residual_edge_conv(point_cloud, repeat=num_edge_conv, k=k,
                                                num_filters=num_edge_filters, trainable=is_training,
                                                debug=debug, _plan=plan)

    E.mark_head_gradients_complete()                               # (backward: the head's gradients are final from here on)
    if nofc:                                                       # model.py:45-58
        last, _, _ = E.as2d(tensors[-1])
        fin = E.conv_bn_act(last, "Final", num_class, relu=True)
        return fin.view(B, N, num_class)

    fcf = ops._listify(num_fc_filters, num_fc, "num_filters")
    # Plane mode (E.HEAD_PLANES): MergedEdgeConv, FC0 and FC1 -- 98 % of the GEMM flops -- read their operands as 16-bit planes
    # written once by the producing pass (csrc/gemm_pl.hip, planes_bn.hip).  The EdgeConv outputs (first ctot - 1024 channels
    # of `big`, and the conv1 copies) are split by one pass each; MergedEdgeConv's BatchNorm pass writes its 1024 channels
    # straight into big's plane set (the fp32 slice of `big` is never written: nothing else reads it -- the global max-pool
    # is taken on the GEMM output) and FC0's writes the planes FC1 reads.
    if pl_head:
        kmax = max(ops._listify(k, num_edge_conv, "k"))
        c.ensure_plane_scales(R * kmax)
        bigP = c.new_planes(R, ctot, "act")
        bigP.cols_view(0, ctot - 1024).fill_from(big[:, :ctot - 1024])
        merged, g = E.conv_bn_act(merged_in, "MergedEdgeConv", 1024, relu=True, out=big[:, ctot - 1024:],
                                  plane_out=bigP.cols_view(ctot - 1024, ctot), f32_out=False, gmax=(B, N))   # model.py:65-77
        c.planes[c.plane_key(big)] = bigP
    else:
        # model.py:65-77: MergedEdgeConv, then the max-pool over the points of each cloud -- taken on the GEMM output (in its
        # epilogue where the tile shape allows) and normalised afterwards: BN + ReLU are non-decreasing
        merged, g = E.conv_bn_act(merged_in, "MergedEdgeConv", 1024, relu=True, out=big[:, ctot - 1024:], gmax=(B, N))
    tensors.append(E.rank4(merged, B, N))                          # model.py:74

    if num_fc < 1:
        # model.py:80-101 with no FC layer: concat([tile(g)] + tensors) -> (dropout) -> Final
        E.tile_rows(g, bigfull[:, :1024], N)                        # model.py:80-81
        net = E.dropout(bigfull, E.DROPOUT_KEEP) if is_training else bigfull      # model.py:90-91
        fin = E.conv_bn_act(net, "Final", num_class, relu=True)    # model.py:94-101
        return fin.view(B, N, num_class)

    # model.py:80-88: concat([tile(g)] + tensors) -> fc.  FC0 is split as described in the docstring.
    with E.variable_scope("FC0"):
        wleaf = c.get_variable("weights", (1024 + ctot, fcf[0]))
    gb = E.plain_gemm(g, wleaf, (0, 1024), fcf[0])                 # per-cloud term (B, fcf0)
    # model.py:90-91: tf.nn.dropout(net, 0.7) behind the LAST fc layer when training -- fused into that layer's BatchNorm passes
    keep = E.DROPOUT_KEEP if is_training else None
    if pl_head and num_fc >= 2:
        net = E.conv_bn_act(big, "FC0", fcf[0], relu=True, gbias=gb, rpg=N, w_rows=(1024, 1024 + ctot, 1024 + ctot),
                            plane_out=c.new_planes(R, fcf[0], "act"), f32_out=False)
    else:
        net = E.conv_bn_act(big, "FC0", fcf[0], relu=True, gbias=gb, rpg=N, w_rows=(1024, 1024 + ctot, 1024 + ctot),
                            drop_keep=keep if num_fc == 1 else None)
    for i in range(1, num_fc):                                     # ops.py:151-160
        net = E.conv_bn_act(net, "FC%d" % i, fcf[i], relu=True, drop_keep=keep if i == num_fc - 1 else None)
    fin = E.conv_bn_act(net, "Final", num_class, relu=True)        # model.py:94-101 (BN + ReLU on the logits)
    return fin.view(B, N, num_class)                                # model.py:104 squeeze
