#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE in the build container.

Run from the repo root:   python tests/golden/make_golden.py

The reference (LouChao98/VLGAE, mounted read-only at /root/reference) has no tests or
golden vectors of its own for this path, so parity is pinned on
outputs of the reference itself, produced here and committed as small .npz files.
Only arrays are written: inputs and the reference's outputs.  No reference source, byte
code or pickled module is stored.  The GPU box never sees /root/reference.

Reference entry points exercised (paths relative to /root/reference):
  * src/model/torch_struct/distributions.py:245-265  DMV1o, DMV1o.merge
  * src/model/torch_struct/distributions.py:116-133,162-174,190-193  .max/.argmax/.marginals/.partition
  * src/model/torch_struct/dmv.py:19-66              DMV1oStruct._dp (inside; outside = autograd)
  * src/model/torch_struct/deptree.py:25-76,213-228  DepTree._dp, DepTree.enumerate
  * src/model/ldndmv.py:185-209                      scorer -> DP glue (gather by token, tril/triu select, function
                                                     mask, root gather, merge), re-issued with the same torch ops
  * src/model/joint.py:406-419                       DependencyBoxRel.gather_logit_simple
  * src/model/joint.py:670-674                       attention-fuse (needs a constructed module, so
                                                     those five lines are re-issued here with the
                                                     same torch ops on the same tensors)
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

warnings.filterwarnings("ignore", message="Named tensors")
torch.set_num_threads(8)

ts = _ref_import.import_torch_struct()
DMV1o, DependencyCRF = ts.DMV1o, ts.DependencyCRF


def _np(t):
    return t.detach().rename(None).cpu().numpy() if t.names and any(t.names) else t.detach().cpu().numpy()


def make_lengths(g, B, L, mode):
    if mode == "full":
        return torch.full((B,), L, dtype=torch.long)
    if mode == "rand":
        ln = torch.randint(1, L + 1, (B,), generator=g)
        ln[0] = L
        return ln
    return torch.tensor(mode, dtype=torch.long)


def dmv_inputs(seed, B, L, normalise_attach):
    """Synthetic potentials shaped like what the scorer emits."""
    g = torch.Generator().manual_seed(seed)
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1)
    attach = torch.randn(B, L, L, 2, generator=g)
    if normalise_attach:
        attach = attach.log_softmax(2)
    root = torch.randn(B, L, generator=g).log_softmax(-1)
    return g, dec, attach, root


def dmv_case(name, seed, B, L, lengths_mode, normalise_attach=False, store_merged=False):
    g, dec, attach, root = dmv_inputs(seed, B, L, normalise_attach)
    lengths = make_lengths(g, B, L, lengths_mode)
    wts = torch.rand(B, generator=g) + 0.5
    out = dict(dec=_np(dec), attach=_np(attach), root=_np(root), lengths=_np(lengths), wts=_np(wts))

    mdec, mattach = DMV1o.merge(dec, attach, root)
    if store_merged:
        out["merged_dec"], out["merged_attach"] = _np(mdec), _np(mattach)

    for tag, dt in (("", torch.float32), ("64", torch.float64)):
        d = mdec.to(dt).detach().requires_grad_()
        a = mattach.to(dt).detach().requires_grad_()
        dist = DMV1o([d, a], lengths)
        logZ = dist.partition                                   # [B,1]
        gd, ga = torch.autograd.grad(logZ.sum(), [d, a])
        out["logZ" + tag], out["grad_dec" + tag], out["grad_attach" + tag] = _np(logZ), _np(gd), _np(ga)

        d = mdec.to(dt).detach().requires_grad_()
        a = mattach.to(dt).detach().requires_grad_()
        dist = DMV1o([d, a], lengths)
        mx = dist.max
        mgd, mga = torch.autograd.grad(mx.sum(), [d, a])
        out["max" + tag], out["maxgrad_dec" + tag], out["maxgrad_attach" + tag] = _np(mx), _np(mgd), _np(mga)

    # weighted upstream gradient (exercises grad_logZ scaling), fp32
    d = mdec.detach().requires_grad_()
    a = mattach.detach().requires_grad_()
    dist = DMV1o([d, a], lengths)
    wgd, wga = torch.autograd.grad((dist.partition.squeeze(-1) * wts).sum(), [d, a])
    out["wgrad_dec"], out["wgrad_attach"] = _np(wgd), _np(wga)

    # lazy properties used by the callers (joint.py:254-258, ldndmv.py:294-303)
    d = mdec.detach().requires_grad_()
    a = mattach.detach().requires_grad_()
    dist = DMV1o([d, a], lengths)
    out["argmax"] = _np(dist.argmax)                            # [B,N,N,2] 0/1
    out["marginals"] = _np(dist.marginals)                      # [B,N,N,2]
    arc = dist.argmax.sum(-1).nonzero()
    predicted = lengths.new_zeros(B, L + 1)
    predicted[arc[:, 0], arc[:, 2]] = arc[:, 1]
    out["predicted"] = _np(predicted)
    arc_margin = torch.from_numpy(out["grad_attach"]).sum(-1)
    out["arc_marginal"] = _np(arc_margin)

    # MBR chain: DependencyCRF over arc marginals, Max semiring (ldndmv.py:294-299)
    crf = DependencyCRF(arc_margin.clone(), lengths)
    out["mbr_argmax"] = _np(crf.argmax)
    out["mbr_max"] = _np(crf.max)

    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: B={B} L={L} lengths={out['lengths'].tolist()[:8]} logZ[0]={out['logZ'][0, 0]:.6f}")


def rules_case(name, seed, B, L, T, lengths_mode, use_mask, root_per_sentence=False):
    """Scorer -> DP glue of DiscriminativeNDMV._forward (src/model/ldndmv.py:185-209), re-issued with the same torch
    ops on synthetic rule tables (the module itself needs a DataModule), followed by the reference DMV1o."""
    LEFT, RIGHT = 0, 1
    g = torch.Generator().manual_seed(seed)
    attach_rule = torch.randn(B, L, T, 2, 2, generator=g).log_softmax(2)
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1)
    root_rule = (torch.randn(B if root_per_sentence else 1, T, generator=g)).log_softmax(-1)
    token = torch.randint(0, T, (B, L), generator=g)
    head_mask = (torch.rand(B, L, generator=g) < 0.15) if use_mask else torch.zeros(B, L, dtype=torch.bool)
    lengths = make_lengths(g, B, L, lengths_mode)
    INF = 1e20                                                    # src/__init__.py:110
    out = dict(attach_rule=_np(attach_rule), dec=_np(dec), root_rule=_np(root_rule), token=_np(token),
               head_mask=_np(head_mask), lengths=_np(lengths), root_per_sentence=np.int64(root_per_sentence))
    for tag, dt in (("", torch.float32), ("64", torch.float64)):
        ar = attach_rule.to(dt).requires_grad_()
        dc = dec.to(dt).requires_grad_()
        rr = root_rule.to(dt).requires_grad_()
        b, n = B, L
        target_size = torch.Size([b, n, n, 2, 2])
        attach_prob = ar.gather(2, token.reshape(b, 1, n, 1, 1).expand(target_size))            # ldndmv.py:189-190
        left_mask = torch.tril(torch.ones(n, n, dtype=dt), diagonal=-1)
        right_mask = torch.triu(torch.ones(n, n, dtype=dt), diagonal=1)
        attach_prob = attach_prob[..., LEFT, :] * left_mask.unsqueeze(0).unsqueeze(-1) \
            + attach_prob[..., RIGHT, :] * right_mask.unsqueeze(0).unsqueeze(-1)                 # :191-194
        if use_mask:
            attach_prob = attach_prob.masked_fill(head_mask.view(b, n, 1, 1), -INF)              # :195-199 (not in place)
        root_prob = rr.expand(b, -1)
        root = torch.gather(root_prob, 1, token)                                                 # :207
        md, ma = DMV1o.merge(dc, attach_prob, root)                                              # :209
        if dt == torch.float64:                # merge builds fp32 buffers (torch.full): redo its five lines in fp64
            N = L + 1
            ma = torch.full((b, N, N, 2), -1e12, dtype=dt)
            md = torch.full((b, N, 2, 2, 2), -1e12, dtype=dt)
            ma[:, 0, 1:, 1] = root
            ma[:, 1:, 1:, :] = attach_prob
            md[:, 0, 1, :, :] = 0
            md[:, 1:] = dc
        dist = DMV1o([md, ma], lengths)
        logZ = dist.partition
        g_ar, g_dc, g_rr = torch.autograd.grad(logZ.sum(), [ar, dc, rr])
        out["logZ" + tag], out["grad_rule" + tag] = _np(logZ), _np(g_ar)
        out["grad_dec" + tag], out["grad_root" + tag] = _np(g_dc), _np(g_rr)
        if tag == "":
            out["merged_dec"], out["merged_attach"] = _np(md), _np(ma)
            arc = DMV1o([md.detach().requires_grad_(), ma.detach().requires_grad_()], lengths).argmax.sum(-1).nonzero()
            predicted = lengths.new_zeros(B, L + 1)
            predicted[arc[:, 0], arc[:, 2]] = arc[:, 1]
            out["predicted"] = _np(predicted)
            out["max"] = _np(DMV1o([md.detach(), ma.detach()], lengths).max)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: B={B} L={L} T={T} logZ[0]={out['logZ'][0, 0]:.6f}")


def deptree_case(name, seed, B, N, lengths_mode, enumerate_=False, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    arc = torch.randn(B, N, N, generator=g) * scale
    lengths = make_lengths(g, B, N - 1, lengths_mode)
    wts = torch.rand(B, generator=g) + 0.5
    out = dict(arc=_np(arc), lengths=_np(lengths), wts=_np(wts))
    for tag, dt in (("", torch.float32), ("64", torch.float64)):
        a = arc.to(dt).clone()
        dist = DependencyCRF(a, lengths)
        out["logZ" + tag] = _np(dist.partition)                 # [B]
        out["marginals" + tag] = _np(dist.marginals)            # [B,N,N]
        out["max" + tag] = _np(dist.max)
        out["argmax" + tag] = _np(dist.argmax)
    a = arc.clone().requires_grad_()
    dist = DependencyCRF(a, lengths)
    (wg,) = torch.autograd.grad((dist.partition * wts).sum(), [a])
    out["wgrad"] = _np(wg)
    if enumerate_:
        # brute-force over all projective single-root trees: the only known-answer device in-tree
        from torch_struct.deptree import DepTree
        from torch_struct import LogSemiring, MaxSemiring
        out["enum_logZ"] = _np(DepTree(LogSemiring).enumerate(arc.double(), non_proj=False, multi_root=False)[0])
        out["enum_max"] = _np(DepTree(MaxSemiring).enumerate(arc.double(), non_proj=False, multi_root=False)[0])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: B={B} N={N} logZ[0]={out['logZ'][0]:.6f}")


def align_cases():
    src, joint = _ref_import.import_joint()
    fn = joint.DependencyBoxRel.gather_logit_simple
    for name, seed, B, A, Q, V, d, bf16 in (
        ("align_B4_A4_Q22_V10_s0", 0, 4, 4, 22, 10, 128, False),
        ("align_B3_A5_Q7_V3_d32_s1", 1, 3, 5, 7, 3, 32, False),
        ("align_B8_A8_Q82_V36_s2_bf16", 2, 8, 8, 82, 36, 128, True),
    ):
        g = torch.Generator().manual_seed(seed)
        txt = torch.randn(B, Q, d, generator=g)
        vis = torch.randn(A, V, d, generator=g)
        if bf16:  # bf16-rounded inputs, reference run in fp32 on the up-cast values
            txt, vis = txt.bfloat16().float(), vis.bfloat16().float()
        tmask = torch.rand(B, Q, generator=g) > 0.1
        vmask = torch.rand(A, V, generator=g) > 0.1
        tmask[:, 0] = False                                     # root slot is masked (joint.py:204,248-249)
        att = fn(None, None,
                 (vis.refine_names("A", "V", "Y"), vmask.refine_names("A", "V"), None),
                 (txt.refine_names("B", "Q", "X"), tmask.refine_names("B", "Q"), None), None)
        assert att.names == ("B", "A", "Q", "V")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), txt=_np(txt), vis=_np(vis),
                            tmask=_np(tmask), vmask=_np(vmask), attmap=_np(att),
                            neg_inf=np.float32(-src.INF))
        print(f"{name}: attmap {tuple(att.shape)}")


def grounding_cases():
    """The reference's own gather_logit_simple -> loss_grounding_factor_ce (joint.py:406-419, 439-491), with
    gradients to both feature tensors through torch autograd.  `self` is a namespace carrying only what the two
    methods read (cfg.loss_grounding_args, vis_factor_names, pos_for_*)."""
    from types import SimpleNamespace as NS
    src, joint = _ref_import.import_joint()
    gather = joint.DependencyBoxRel.gather_logit_simple
    loss_fn = joint.DependencyBoxRel.loss_grounding_factor_ce
    for name, seed, B, L, split, d, prior, w_v2t, n_tag in (
        ("ground_B4_L10_obj6_s0", 0, 4, 10, (("obj", 6),), 32, True, 1.0, 9),
        ("ground_B5_L7_obj5_rel25_attr5_img1_s1", 1, 5, 7, (("obj", 5), ("rel", 25), ("attr", 5), ("img", 1)), 64, True, 1.0, 12),
        ("ground_B3_L6_obj4_rel16_s2_noprior", 2, 3, 6, (("obj", 4), ("rel", 16)), 32, False, 0.0, 8),
        ("ground_B8_L40_obj36_s3", 3, 8, 40, (("obj", 36),), 128, True, 1.0, 45),
    ):
        g = torch.Generator().manual_seed(seed)
        names, widths = [n for n, _ in split], [w for _, w in split]
        V, Q = sum(widths), 2 * (L + 1)
        lengths = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
        lengths[0] = L
        wmask = torch.arange(L)[None] < lengths[:, None]                       # vp.mask [B,L]
        m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), wmask], 1)
        tmask = torch.cat([m1, m1], 1)                                          # joint.py:248-249
        vmask = torch.rand(B, V, generator=g) > 0.15
        vmask[:, 0] = True
        txt = (torch.randn(B, Q, d, generator=g) * 0.5).requires_grad_(True)
        vis = (torch.randn(B, V, d, generator=g) * 0.5).requires_grad_(True)
        marg = torch.rand(B, Q, generator=g) * tmask                            # txt_marginal: constants (joint.py:251-268)
        tag = torch.randint(0, n_tag, (B, L), generator=g)
        pos = dict(obj=torch.tensor([0, 1, 2]), rel=torch.tensor([2, 3]), attr=torch.tensor([4]))   # 2 is in two sets
        me = NS(cfg=NS(loss_grounding_args=NS(use_pos_prior=prior, vis2txt=w_v2t)), vis_factor_names=names,
                pos_for_obj=pos["obj"], pos_for_rel=pos["rel"], pos_for_attr=pos["attr"])
        vis_p = (vis.refine_names("A", "V", "Y"), vmask.refine_names("A", "V"), widths)
        txt_p = (txt.refine_names("B", "Q", "X"), tmask.refine_names("B", "Q"), marg)
        logit = gather(me, None, vis_p, txt_p, None)
        num_token = int(lengths.sum())
        total, parts = loss_fn(me, {"match_logit": logit, "txt_packed": txt_p, "vis_packed": vis_p},
                               NS(tag=tag, num_token=num_token))
        # the raw sums (the reported losses are s / (s.detach() + 1e-6) * num_token, i.e. ~num_token by construction)
        with torch.no_grad():
            att = logit.rename(None).clone()    # the prior was already subtracted in place on the diagonal pairs
            mv = att.max(3).values.log_softmax(1)
            t2v = -(mv.diagonal().T * marg).sum()
            v2t = -(att.max(2).values.log_softmax(0).diagonal().T * vmask).sum()
        g_txt, g_vis = torch.autograd.grad(total, [txt, vis])
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), txt=_np(txt), vis=_np(vis), tmask=_np(tmask), vmask=_np(vmask),
            marginal=_np(marg), tag=_np(tag), lengths=_np(lengths), num_token=np.int64(num_token),
            factor_names=np.array(names), vis_split=np.array(widths, dtype=np.int64), use_pos_prior=np.bool_(prior),
            vis2txt_weight=np.float32(w_v2t), pos_for_obj=_np(pos["obj"]), pos_for_rel=_np(pos["rel"]),
            pos_for_attr=_np(pos["attr"]), total=_np(total), txt2vis_raw=_np(t2v), vis2txt_raw=_np(v2t),
            loss_txt2vis=_np(parts["txt2vis"]), loss_vis2txt=_np(parts.get("mt_vis2txt", torch.zeros(()))),
            attmap_prior=_np(att), g_txt=_np(g_txt), g_vis=_np(g_vis), neg_inf=np.float32(-src.INF))
        print(f"{name}: total {float(total):.4f} txt2vis_raw {float(t2v):.4f} vis2txt_raw {float(v2t):.4f}")


def reduced_cases():
    """The reference's own gather_logit_reduced -> loss_grounding_cap_img_ll (joint.py:421-432, 493-499: cross-entropy over the
    images of every caption) with torch autograd gradients to both feature tensors."""
    from types import SimpleNamespace as NS
    src, joint = _ref_import.import_joint()
    cls = joint.DependencyBoxRel
    for name, seed, B, L, V, d in (("reduced_B4_L6_V7_s0", 0, 4, 6, 7, 32), ("reduced_B6_L40_V36_s1", 1, 6, 40, 36, 128)):
        g = torch.Generator().manual_seed(seed)
        Q = 2 * (L + 1)
        lengths = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
        lengths[0] = L
        wmask = torch.arange(L)[None] < lengths[:, None]
        m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), wmask], 1)
        tmask = torch.cat([m1, m1], 1)
        vmask = torch.rand(B, V, generator=g) > 0.15
        vmask[:, 0] = True
        txt = (torch.randn(B, Q, d, generator=g) * 0.5).requires_grad_(True)
        vis = (torch.randn(B, V, d, generator=g) * 0.5).requires_grad_(True)
        marg = torch.rand(B, Q, generator=g) * tmask
        me = NS(training=True, criteria=torch.nn.CrossEntropyLoss())
        me.gather_logit_simple = lambda inputs, vis_p, txt_p, vp: cls.gather_logit_simple(me, inputs, vis_p, txt_p, vp)
        vis_p = (vis.refine_names("A", "V", "Y"), vmask.refine_names("A", "V"), [V])
        txt_p = (txt.refine_names("B", "Q", "X"), tmask.refine_names("B", "Q"), marg)
        logit = cls.gather_logit_reduced(me, None, vis_p, txt_p, None)
        loss, _ = cls.loss_grounding_cap_img_ll(me, {"match_logit": logit.rename(None)}, NS(batch_size=B))
        g_logit, = torch.autograd.grad(loss, logit, retain_graph=True)
        g_txt, g_vis = torch.autograd.grad(loss, [txt, vis])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), txt=_np(txt), vis=_np(vis), tmask=_np(tmask), vmask=_np(vmask),
                            marginal=_np(marg), logit=_np(logit.rename(None)), loss=_np(loss), g_logit=_np(g_logit.rename(None)),
                            g_txt=_np(g_txt), g_vis=_np(g_vis))
        print(f"{name}: logit {tuple(logit.shape)} loss {float(loss):.4f}")


def decode_cases():
    """The reference's own gather_logit_simple -> decode_grounding_on_factor (joint.py:406-419, 512-629) through a stub
    `self` / `vp` carrying only what the method reads.  Saved: the inputs, the diagonal block before and after the method's
    in-place edits (the method edits a view of match_logit's diagonal, so it is read back from there), max over V, the
    top-5 VALUES per row (indices of equal values are not defined by torch.argsort) and the method's two result lists."""
    import json
    from types import SimpleNamespace as NS
    src, joint = _ref_import.import_joint()
    gather = joint.DependencyBoxRel.gather_logit_simple
    decode = joint.DependencyBoxRel.decode_grounding_on_factor

    class VP(dict):            # `"vis_box_index" in vp` and attribute access, like the reference's variable pool
        __getattr__ = dict.__getitem__

    for name, seed, B, L, split, d, prior, heur, n_tag, with_index in (
        ("gdecode_B4_L6_obj5_s0", 0, 4, 6, (("obj", 5),), 32, True, True, 6, False),
        ("gdecode_B5_L7_obj4_rel16_attr4_img1_s1", 1, 5, 7, (("obj", 4), ("rel", 16), ("attr", 4), ("img", 1)), 32, True, True, 7, True),
        ("gdecode_B3_L5_obj6_rel36_s2_noprior", 2, 3, 5, (("obj", 6), ("rel", 36)), 32, False, True, 6, False),
        ("gdecode_B3_L9_obj8_attr8_s3_noheur", 3, 3, 9, (("obj", 8), ("attr", 8)), 64, True, False, 6, True),
    ):
        g = torch.Generator().manual_seed(seed)
        names, widths = [n for n, _ in split], [w for _, w in split]
        V, Q = sum(widths), 2 * (L + 1)
        lengths = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
        lengths[0] = L
        wmask = torch.arange(L)[None] < lengths[:, None]
        m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), wmask], 1)
        tmask = torch.cat([m1, m1], 1)
        vmask = torch.rand(B, V, generator=g) > 0.1
        vmask[:, 0] = True
        txt = torch.randn(B, Q, d, generator=g) * 0.5
        vis = torch.randn(B, V, d, generator=g) * 0.5
        tag = torch.randint(0, n_tag, (B, L), generator=g)
        pos = dict(obj=torch.tensor([0, 1, 2]), rel=torch.tensor([2, 3]), attr=torch.tensor([4]))
        box_index = [torch.randperm(200, generator=g)[:widths[0]].tolist() for _ in range(B)] if with_index else None
        me = NS(cfg=NS(decode_grounding_args=NS(use_pos_prior=prior, use_heuristic=heur)), vis_factor_names=names,
                pos_for_obj=pos["obj"], pos_for_rel=pos["rel"], pos_for_attr=pos["attr"])
        vis_p = (vis.refine_names("A", "V", "Y"), vmask.refine_names("A", "V"), widths)
        txt_p = (txt.refine_names("B", "Q", "X"), tmask.refine_names("B", "Q"), None)
        logit = gather(me, None, vis_p, txt_p, None)
        plain = logit.rename(None)
        diag_before = plain.diagonal().permute(2, 0, 1).clone()              # [B,Q,V]
        max_v = plain.max(3).values.clone()                                   # [B,A,Q]
        vp = VP(tag=tag, mask=wmask, seq_len_cpu=lengths.tolist())
        if box_index is not None:
            vp["vis_box_index"] = torch.tensor(box_index)
        out = decode(me, {"match_logit": logit, "vis_packed": vis_p, "txt_packed": txt_p}, vp)
        diag_after = plain.diagonal().permute(2, 0, 1).clone()               # the method's in-place edits land here
        top_vals = diag_after.sort(-1, descending=True).values[..., :5]
        to_img = [[int(t) for t in row] for row in out["txt_to_img"]]
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), txt=_np(txt), vis=_np(vis), tmask=_np(tmask), vmask=_np(vmask), tag=_np(tag),
            lengths=_np(lengths), factor_names=np.array(names), vis_split=np.array(widths, dtype=np.int64),
            use_pos_prior=np.bool_(prior), use_heuristic=np.bool_(heur), pos_for_obj=_np(pos["obj"]),
            pos_for_rel=_np(pos["rel"]), pos_for_attr=_np(pos["attr"]),
            vis_box_index=np.array(box_index if box_index is not None else [], dtype=np.int64),
            diag_before=_np(diag_before), diag_after=_np(diag_after), max_v=_np(max_v), top_vals=_np(top_vals),
            txt_to_factor=np.array(json.dumps(out["txt_to_factor"])), txt_to_img=np.array(json.dumps(to_img)))
        print(f"{name}: diag {tuple(diag_after.shape)} edited {int((diag_after != diag_before).sum())} entries")


def attnfuse_cases():
    """joint.py:670-674 re-issued with the same torch ops (module construction needs a DataModule)."""
    for name, seed, B, L, V, d, h in (
        ("attnfuse_B4_L10_V10_s0", 0, 4, 10, 10, 128, 256),
        ("attnfuse_B8_L40_V36_s1", 1, 8, 40, 36, 128, 256),
        ("attnfuse_B3_L5_V35_d64_h96_s2", 2, 3, 5, 35, 64, 96),
    ):
        g = torch.Generator().manual_seed(seed)
        vis = torch.randn(B, V, d, generator=g)                 # vis[0]
        txt = torch.randn(B, L + 1, d, generator=g)             # txt[0] (root slot first)
        vis_mid = torch.randn(B, V, h, generator=g)             # vis[3]
        enc_x = torch.randn(B, L, h, generator=g)               # encoded['x']
        ln = torch.nn.LayerNorm(h)
        with torch.no_grad():
            ln.weight.copy_(torch.rand(h, generator=g) + 0.5)
            ln.bias.copy_(torch.randn(h, generator=g) * 0.1)
        dout = torch.randn(B, L, h, generator=g)               # cotangent of encoded['x'] (drawn last: earlier draws unchanged)
        leaves = [vis, txt, vis_mid, enc_x]
        for t_ in leaves:
            t_.requires_grad_(True)
        attmap = torch.einsum("bvd, bqd -> bqv", vis, txt[:, 1:]).softmax(2)
        x = torch.einsum("bqv,bvh->bqh", attmap, vis_mid)
        out = ln(enc_x + x)
        grads = torch.autograd.grad(out, leaves + [ln.weight, ln.bias], dout)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), vis=_np(vis), txt=_np(txt), vis_mid=_np(vis_mid),
                            enc_x=_np(enc_x), ln_weight=_np(ln.weight), ln_bias=_np(ln.bias),
                            ln_eps=np.float32(ln.eps), attmap=_np(attmap), out=_np(out), dout=_np(dout),
                            g_vis=_np(grads[0]), g_txt=_np(grads[1]), g_vis_mid=_np(grads[2]), g_enc_x=_np(grads[3]),
                            g_ln_weight=_np(grads[4]), g_ln_bias=_np(grads[5]))
        print(f"{name}: out {tuple(out.shape)}")


def arcenc_cases():
    """joint.py:281-287 re-issued with the same torch ops (the method around them needs the whole model).
    The X = 128 case keeps the fixture small: w1 is stored as rank-4 factors (w1 = sum_r u[:,r] v[:,r] z[:,r], rebuilt by
    the tests) and only a strided sample of its gradient is kept."""
    for name, seed, B, C, X, rank in (("arcenc_B3_C6_X32_s0", 0, 3, 6, 32, 0), ("arcenc_B4_C11_X64_s1", 1, 4, 11, 64, 0),
                                      ("arcenc_B2_C41_X128_s2", 2, 2, 41, 128, 4)):
        g = torch.Generator().manual_seed(seed)
        child = (torch.randn(B, C, X, generator=g) * 0.5).requires_grad_(True)
        parent = (torch.randn(B, C, X, generator=g) * 0.5).requires_grad_(True)
        extra = {}
        if rank:
            u, v, z = (torch.randn(X, rank, generator=g) * 0.3 for _ in range(3))
            w1 = torch.einsum("xr,hr,yr->xhy", u, v, z).contiguous().requires_grad_(True)
            extra = dict(w1_u=_np(u), w1_v=_np(v), w1_z=_np(z))
        else:
            w1 = (torch.randn(X, X, X, generator=g) * (1.0 / X)).requires_grad_(True)   # arc_encoder_w1
        w2 = (torch.randn(X, X, generator=g) * (1.0 / X ** 0.5)).requires_grad_(True)   # arc_encoder_w2
        b = (torch.randn(X, generator=g) * 0.1).requires_grad_(True)                    # arc_encoder_b
        dout = torch.randn(B, C, X, generator=g)
        arc = torch.einsum("bcx,xhy,bcy->bch", child, w1, parent) + torch.matmul(child + parent, w2) + b
        grads = torch.autograd.grad(arc, [child, parent, w1, w2, b], dout)
        if rank:
            extra["g_w1_sample"] = _np(grads[2][::5, ::7, ::3])
        else:
            extra.update(w1=_np(w1), g_w1=_np(grads[2]))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), child=_np(child), parent=_np(parent), w2=_np(w2),
                            b=_np(b), dout=_np(dout), arc=_np(arc), g_child=_np(grads[0]), g_parent=_np(grads[1]),
                            g_w2=_np(grads[3]), g_b=_np(grads[4]), **extra)
        print(f"{name}: arc {tuple(arc.shape)}")


def boxrel_cases():
    """VisBoxRelSimpleEncoder.forward (vis_encoder/box_rel.py:29-52), the reference's own module executed: pairwise-mean
    input -> rel_fc (Linear + LeakyReLU), with gradients to the region features and the rel_fc parameters."""
    import _ref_import
    _ref_import.import_joint()
    from src.model.vis_encoder import VisBoxRelSimpleEncoder
    for name, seed, B, R, n_in, n_hidden in (("boxrel_B3_R5_in48_h32_s0", 0, 3, 5, 48, 32), ("boxrel_B1_R35_in64_h64_s1", 1, 1, 35, 64, 64)):
        torch.manual_seed(seed)
        enc = VisBoxRelSimpleEncoder(n_in=n_in, n_hidden=n_hidden, dropout=0., activate=True, use_attr=True, use_img=False,
                                     img_feat=True)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            enc.rel_fc.linear.bias.copy_(torch.randn(n_hidden, generator=g) * 0.1)     # reset_parameters zeroes it
        feat = torch.randn(B, R, n_in, generator=g).requires_grad_(True)
        out = enc({"vis_box_feat": feat}, None)
        dout = torch.randn(B, R * R, n_hidden, generator=g)
        w, b = enc.rel_fc.linear.weight, enc.rel_fc.linear.bias
        grads = torch.autograd.grad(out["rel"], [feat, w, b], dout)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), feat=_np(feat), weight=_np(w), bias=_np(b), rel=_np(out["rel"]),
                            dout=_np(dout), g_feat=_np(grads[0]), g_weight=_np(grads[1]), g_bias=_np(grads[2]),
                            slope=np.float32(enc.rel_fc.activation.negative_slope))
        print(f"{name}: rel {tuple(out['rel'].shape)}")


def langfeat_cases():
    """The reference's own `DependencyBoxRel.lang_feat_max_tree` (joint.py:235-292) executed as an unbound method on a
    namespace that carries what it reads: the three encoders (the reference's `MLP`, nn/common.py:23-51: word encoder
    without activation, child / parent with LeakyReLU, config/model/vlgae.yaml:69-73 + joint.py:216-222; dropout 0), the arc
    encoder's parameters and cfg.add_marginal; `DMV1o` inside it is the reference's.  Gradients of <txt, dout> by torch
    autograd.  The d = 128 case stores w1 as rank-4 factors and a strided sample of its gradient (like arcenc_cases)."""
    from types import SimpleNamespace as NS
    src, joint = _ref_import.import_joint()
    from src.model.nn import MLP
    fn = joint.DependencyBoxRel.lang_feat_max_tree
    for name, seed, B, L, h, d, rank, add_marginal in (("langfeat_B3_L6_h64_d32_s0", 0, 3, 6, 64, 32, 0, True),
                                                       ("langfeat_B4_L9_h128_d64_s1_nomarg", 1, 4, 9, 128, 64, 4, False),
                                                       ("langfeat_B2_L40_h256_d128_s2", 2, 2, 40, 256, 128, 4, True)):
        torch.manual_seed(seed)
        g = torch.Generator().manual_seed(seed)
        enc = {k: MLP(n_in=h, n_hidden=d, dropout=0, activate=(k != "word")) for k in ("word", "child", "parent")}
        with torch.no_grad():
            for m in enc.values():                                                  # reset_parameters zeroes the bias
                m.linear.bias.copy_(torch.randn(d, generator=g) * 0.1)
        extra = {}
        if rank:
            u, v, z = (torch.randn(d, rank, generator=g) * 0.3 for _ in range(3))
            w1 = torch.nn.Parameter(torch.einsum("xr,hr,yr->xhy", u, v, z).contiguous())
            extra = dict(w1_u=_np(u), w1_v=_np(v), w1_z=_np(z))
        else:
            w1 = torch.nn.Parameter(torch.randn(d, d, d, generator=g) * (1.0 / d))
        w2 = torch.nn.Parameter(torch.randn(d, d, generator=g) * (1.0 / d ** 0.5))
        b_arc = torch.nn.Parameter(torch.randn(d, generator=g) * 0.1)
        me = NS(cfg=NS(add_marginal=add_marginal), word_encoder=enc["word"], child_encoder=enc["child"],
                parent_encoder=enc["parent"], arc_encoder_w1=w1, arc_encoder_w2=w2, arc_encoder_b=b_arc)
        lengths = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
        lengths[0] = L
        mask = torch.arange(L)[None] < lengths[:, None]
        x = (torch.randn(B, L, h, generator=g) * 0.5).requires_grad_(True)
        _, dec, attach, root = dmv_inputs(seed + 100, B, L, False)
        mdec, mattach = ts.DMV1o.merge(dec, attach, root)
        vp = NS(mask=mask, seq_len=lengths, batch_size=B, max_len=L)
        txt, txt_mask, txt_marginal = fn(me, None, {"x": x}, {"merged_dec": mdec, "merged_attach": mattach}, vp)
        txt = txt.rename(None)
        dout = torch.randn(B, 2 * (L + 1), d, generator=g) * txt_mask.rename(None).unsqueeze(-1)   # masked rows never reach the loss
        params = [enc["word"].linear.weight, enc["word"].linear.bias, enc["child"].linear.weight, enc["child"].linear.bias,
                  enc["parent"].linear.weight, enc["parent"].linear.bias, w1, w2, b_arc]
        grads = torch.autograd.grad(txt, [x] + params, dout)
        # `predicted`, joint.py:252-258 (the method keeps it local): the same three lines on the same potentials
        arc = ts.DMV1o([mdec.detach().requires_grad_(), mattach.detach().requires_grad_()], lengths).argmax.sum(-1).nonzero()
        predicted = torch.zeros(B, L + 1, dtype=torch.long)
        predicted[arc[:, 0], arc[:, 2]] = arc[:, 1]
        if rank:
            extra["g_w1_sample"] = _np(grads[7][::5, ::7, ::3])
        else:
            extra.update(w1=_np(w1), g_w1=_np(grads[7]))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), x=_np(x), lengths=_np(lengths), merged_dec=_np(mdec), merged_attach=_np(mattach),
            w_word=_np(params[0]), b_word=_np(params[1]), w_child=_np(params[2]), b_child=_np(params[3]), w_parent=_np(params[4]),
            b_parent=_np(params[5]), w2=_np(w2), b_arc=_np(b_arc), add_marginal=np.bool_(add_marginal),
            slope=np.float32(enc["child"].activation.negative_slope), predicted=_np(predicted), txt=_np(txt),
            txt_mask=_np(txt_mask.rename(None)), txt_marginal=_np(txt_marginal.rename(None)), dout=_np(dout), g_x=_np(grads[0]),
            g_w_word=_np(grads[1]), g_b_word=_np(grads[2]), g_w_child=_np(grads[3]), g_b_child=_np(grads[4]),
            g_w_parent=_np(grads[5]), g_b_parent=_np(grads[6]), g_w2=_np(grads[8]), g_b_arc=_np(grads[9]), **extra)
        print(f"{name}: txt {tuple(txt.shape)} marginal sum {float(txt_marginal.rename(None).sum()):.4f}")


def scorer_cases():
    """The reference's own `DiscriminativeNDMV._forward` (ldndmv.py:171-216) executed as an unbound method on a namespace
    carrying the reference's modules (MLP feed-forwards, DMVSkipConnectEncoder, three DMVFactorizedBilinear scorers) --
    context_mode 'none' (extract_sent_repr / construct_token_repr pass the embedding through), no pretrained-DMV init.
    The scorers' projected inputs are captured with forward hooks; gradients of <merged potentials, cotangents> w.r.t. them
    (and w.r.t. root_rule) by torch autograd through the reference's ops."""
    from types import SimpleNamespace as NS
    src, joint = _ref_import.import_joint()
    from src.model import ldndmv
    from src.model.nn import MLP, DMVSkipConnectEncoder, DMVFactorizedBilinear
    fn = ldndmv.DiscriminativeNDMV._forward
    src.trainer = NS(current_epoch=100)
    for name, seed, B, L, T, H, r, use_mask in (("scorer_B3_L7_T9_H24_r8_s0", 0, 3, 7, 9, 24, 8, True),
                                                ("scorer_B4_L12_T45_H32_r16_s1", 1, 4, 12, 45, 32, 16, False),
                                                ("scorer_B2_L40_T45_H32_r16_s2", 2, 2, 40, 45, 32, 16, True)):
        torch.manual_seed(seed)
        g = torch.Generator().manual_seed(seed)
        E = 12
        me = NS(cfg=NS(extended_valence=True, function_mask=use_mask, init_epoch=0), dmv=None,
                head_ff=MLP(n_in=E, n_hidden=H), child_ff=MLP(n_in=E, n_hidden=H), root_ff=MLP(n_in=10, n_hidden=H),
                dec_ff=MLP(n_in=10, n_hidden=H), mid_ff=DMVSkipConnectEncoder(hidden_size=H, n_bottleneck=6),
                attach_scorer=DMVFactorizedBilinear(n_in=H, r=r), dec_scorer=DMVFactorizedBilinear(n_in=H, r=r),
                root_scorer=DMVFactorizedBilinear(n_in=H, r=r), token_emb=torch.randn(T, E, generator=g),
                root_emb=torch.randn(1, 10, generator=g), dec_emb=torch.randn(2, 10, generator=g),
                function_mask=torch.tensor([1, 4]), extract_sent_repr=lambda enc: (None, None),
                construct_token_repr=lambda emb, ctx, vp: emb)
        cap = {}
        hooks = [m.register_forward_hook(lambda mod, inp, out, k=k: cap.__setitem__(k, out))
                 for k, m in (("x1", me.attach_scorer.project1), ("x2", me.attach_scorer.project2), ("y1", me.dec_scorer.project1),
                              ("y2", me.dec_scorer.project2))]
        token = torch.randint(0, T, (B, L), generator=g)
        tag = torch.randint(0, 6, (B, L), generator=g)
        emb = torch.randn(B, L, E, generator=g)
        out = fn(me, {"token": token, "tag": tag}, {"emb": emb}, NS(batch_size=B, max_len=L))
        for h in hooks:
            h.remove()
        md, ma = out["merged_dec"], out["merged_attach"]
        g_md = torch.rand(B, L + 1, 2, 2, 2, generator=g)
        g_ma = torch.rand(B, L + 1, L + 1, 2, generator=g)
        grads = torch.autograd.grad([md, ma], [cap["x1"], cap["x2"], cap["y1"], cap["y2"], out["root_rule"]], [g_md, g_ma])
        head_mask = tag.unsqueeze(-1).eq(me.function_mask.view(1, 1, -1)).any(-1) if use_mask else torch.zeros(B, L, dtype=torch.bool)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), x1=_np(cap["x1"]), x2=_np(cap["x2"][0]), y1=_np(cap["y1"]), y2=_np(cap["y2"][0]),
            root_rule=_np(out["root_rule"][0]), token=_np(token), head_mask=_np(head_mask), mask_fill=np.float32(-src.INF),
            merged_dec=_np(md), merged_attach=_np(ma), g_merged_dec=_np(g_md), g_merged_attach=_np(g_ma), g_x1=_np(grads[0]),
            g_x2=_np(grads[1][0]), g_y1=_np(grads[2]), g_y2=_np(grads[3][0]), g_root_rule=_np(grads[4].sum(0)))
        print(f"{name}: merged_attach {tuple(ma.shape)} finite min {float(ma[ma > -1e11].min()):.3f}")



def trainstep_cases():
    """One whole training step of the shipped `vlgae` model AS THE REFERENCE WIRES IT, executed by the reference's own methods
    called unbound on namespaces that carry what they read (the modules are the reference's `MLP`, `DMVSkipConnectEncoder`,
    `DMVFactorizedBilinear`, `SharedDropout`; `VarPool`, `reduce_loss` and `DMV1o` are the reference's too):

      JointModelBase.forward order (base.py:215-241), starting at the FROZEN features (round 5):
        VisBoxRelSimpleEncoder.forward   vis_encoder/box_rel.py:29-52   base.py:229: the reference's own module on vis_box_feat [B,R,n]
                                                            (img_feat, use_attr as the case has the factor, dropout 0, use_img False: vlgae.yaml:27-35)
        MLPEncoder.forward               text_encoder/mlp_encoder.py:36-40   base.py:68 (ModelBase.forward): called unbound on a namespace with
                                                            `linear` = nn.Linear(E, h, bias=False), `shared_dropout` = Identity (shared_dropout: 0,
                                                            vlgae.yaml:21-25) and `dropout` = a recorder that multiplies by a mask drawn from this
                                                            script's generator ((rand >= p) / (1 - p), what nn.Dropout(p) does in training): the mask
                                                            is an input of the fixture (`enc_drop_mask`)
        DependencyBoxRel._forward        joint.py:658-675   feat_fuse_attention (a pass-through copy, :362-398), vis_feat_unprune with
                                                            return_mid (:137-178), lang_feat_word_only (:193-211), the attention fuse
                                                            (:670-674) into a COPY of `encoded` (replace: false), then
        DiscriminativeNDMV._forward      ldndmv.py:171-216  on that copy: context_mode 'mean' reads the FUSED x (extract_sent_repr :226),
                                                            head_ff / mid_ff / scorers -> merged potentials
        DependencyBoxRel._vis_forward    joint.py:677-691   vis_feat_unprune again, lang_feat_max_tree (:235-292) on the caller's
                                                            `encoded` -- i.e. the UN-fused x --, gather_logit_simple (:406-419)
      DependencyBoxRel.loss              joint.py:693-711   DiscriminativeNDMV.loss (ldndmv.py:260-285, viterbi_training: -max.sum()),
                                                            loss_grounding_factor_ce (:439-491; use_pos_prior, vis2txt = 1),
                                                            alpha * mt + (1 - alpha) * dep, alpha = grounding_interpolation = 0.5
      reduce_loss('token')               utility/fn.py:50-56, pipeline.py:124,249-250

    Training mode: the word / child / parent encoders' SharedDropout (p = 0.33, config/model/vlgae.yaml:69-73) is live; its
    `get_mask` is replaced by a recorder that draws from this script's generator so that the masks are inputs of the fixture
    (call order: word encoder in lang_feat_word_only; word, child, parent encoders in lang_feat_max_tree).  The scorer's
    feed-forwards run with dropout 0 (out of the hot path; their dropout is torch's).  Stored: every input, every parameter,
    the dropout masks, intermediate values (the encoders' outputs, fused x, potentials, heads, txt, marginal, the loss terms) and the
    gradient of the reduced loss w.r.t. every input feature (the raw E-d embeddings and n-d region features) and parameter (the two
    encoders' included; `rel_fc` / `attr_fc` of a case whose model has no such factor get no gradient: listed in `unused`).  `w1` of the d = 128 case is stored as rank-4 factors and only a
    strided sample of its gradient is kept."""
    from functools import partial
    from types import SimpleNamespace as NS
    src, joint = _ref_import.import_joint()
    from src.model import ldndmv
    from src.model.nn import MLP, DMVSkipConnectEncoder, DMVFactorizedBilinear
    from src.model.nn.dropout import SharedDropout
    from src.model.text_encoder.mlp_encoder import MLPEncoder
    from src.model.vis_encoder import VisBoxRelSimpleEncoder
    from src.utility.var_pool import VarPool
    from src.utility.fn import reduce_loss
    JB, ND = joint.DependencyBoxRel, ldndmv.DiscriminativeNDMV
    src.trainer = NS(current_epoch=100)
    real_get_mask = SharedDropout.get_mask
    for name, seed, B, L, boxes, factors, h, d, E, Et, T, H, nb, r, rank, p_drop, sc_gain, n_vis in (
            ("trainstep_B3_L6_box4_rel_attr_h64_d32_s0", 0, 3, 6, 4, ("rel", "attr"), 64, 32, 16, 10, 9, 24, 6, 8, 0, 0.33, 6.0, 24),
            ("trainstep_B4_L9_box5_rel_attr_img_h64_d32_s1_nodrop", 1, 4, 9, 5, ("rel", "attr", "img"), 64, 32, 16, 10, 11, 24, 0, 8, 0, 0.0, 8.0, 40),
            ("trainstep_B8_L40_box36_h256_d128_s2", 2, 8, 40, 36, (), 256, 128, 40, 24, 45, 96, 40, 16, 4, 0.33, 10.0, 96)):
        torch.manual_seed(seed)
        g = torch.Generator().manual_seed(seed)
        rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
        # ---- the model: reference modules on two namespaces ----
        enc = {k: MLP(n_in=h, n_hidden=d, dropout=p_drop, activate=(k != "word")) for k in ("word", "child", "parent")}
        with torch.no_grad():
            for m in enc.values():
                m.linear.bias.copy_(rnd(d, sc=0.1))
        if rank:
            u, v, z = (rnd(d, rank, sc=0.3) for _ in range(3))
            w1 = torch.nn.Parameter(torch.einsum("xr,hr,yr->xhy", u, v, z).contiguous())
        else:
            w1 = torch.nn.Parameter(rnd(d, d, d, sc=1.0 / d))
        w2 = torch.nn.Parameter(rnd(d, d, sc=d ** -0.5))
        b_arc = torch.nn.Parameter(rnd(d, sc=0.1))
        pre_match = torch.nn.Linear(h, d, bias=False)
        # the two trainable encoders in front of everything (base.py:229,68): the reference's VisBoxRelSimpleEncoder as a module, and
        # MLPEncoder.forward unbound on a namespace (its __init__ wants an Embedding object; forward reads three attributes)
        venc = VisBoxRelSimpleEncoder(n_in=n_vis, n_hidden=h, dropout=0., activate=True, use_attr="attr" in factors, use_img=False, img_feat=True)
        text_linear = torch.nn.Linear(E, h, bias=False)
        with torch.no_grad():
            for m in (venc.box_fc, venc.rel_fc) + ((venc.attr_fc,) if "attr" in factors else ()):
                m.linear.bias.copy_(rnd(h, sc=0.1))                              # (reset_parameters zeroes it)
            text_linear.weight.copy_(rnd(h, E, sc=E ** -0.5))
        ln = torch.nn.LayerNorm(h)
        with torch.no_grad():
            ln.weight.copy_(torch.rand(h, generator=g) + 0.5)
            ln.bias.copy_(rnd(h, sc=0.1))
        dep = NS(cfg=NS(extended_valence=True, function_mask=False, init_epoch=0, viterbi_training=True, context_mode="mean",
                        variational_mode="none"), dmv=None, variational_enc=None, training=True,
                 head_ff=MLP(n_in=E + h, n_hidden=H), child_ff=MLP(n_in=Et, n_hidden=H), root_ff=MLP(n_in=10, n_hidden=H),
                 dec_ff=MLP(n_in=10, n_hidden=H), mid_ff=DMVSkipConnectEncoder(hidden_size=H, n_bottleneck=nb),
                 attach_scorer=DMVFactorizedBilinear(n_in=H, r=r), dec_scorer=DMVFactorizedBilinear(n_in=H, r=r),
                 root_scorer=DMVFactorizedBilinear(n_in=H, r=r), token_emb=torch.nn.Parameter(rnd(T, Et)),
                 root_emb=torch.nn.Parameter(rnd(1, 10)), dec_emb=torch.nn.Parameter(rnd(2, 10)))
        for fn_name in ("extract_sent_repr", "construct_token_repr", "_forward", "loss"):
            setattr(dep, fn_name, partial(getattr(ND, fn_name), dep))
        # the scorers' default init gives near-uniform rule distributions (scores of ~0.03 around log(1/T)): every tree then scores within
        # 1e-3 of every other and the Viterbi tree is decided by fp32 rounding.  Scale the projections so that scores have a spread of
        # ~1.5 nats, as a trained parser's do.
        with torch.no_grad():
            for sc_mod in (dep.attach_scorer, dep.dec_scorer, dep.root_scorer):
                for p_ in sc_mod.parameters():
                    p_.mul_(sc_gain)
        pos = dict(obj=torch.tensor([0, 1, 2]), rel=torch.tensor([2, 3]), attr=torch.tensor([4]))
        me = NS(cfg=NS(feat_fuse_mode="attention", feat_fuse_args=NS(aug_with_matching=True, replace=False), add_rel="rel" in factors,
                       add_attr="attr" in factors, add_image="img" in factors, add_marginal=True, grounding_interpolation=0.5,
                       loss_grounding_args=NS(use_pos_prior=True, vis2txt=1.0)), training=True, vis_factor_names=["obj", *factors],
                pos_for_obj=pos["obj"], pos_for_rel=pos["rel"], pos_for_attr=pos["attr"], word_encoder=enc["word"],
                child_encoder=enc["child"], parent_encoder=enc["parent"], arc_encoder_w1=w1, arc_encoder_w2=w2, arc_encoder_b=b_arc,
                vis_mlp_pre_matching=pre_match, feat_layernorm=ln, dependency=dep)
        for attr_, fn_name in (("feat_fuse", "feat_fuse_attention"), ("vis_feat", "vis_feat_unprune"), ("lang_feat", "lang_feat_max_tree"),
                               ("lang_feat_word_only", "lang_feat_word_only"), ("gather_logit", "gather_logit_simple"),
                               ("loss_grounding", "loss_grounding_factor_ce")):
            setattr(me, attr_, partial(getattr(JB, fn_name), me))
        # ---- the batch ----
        lengths = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
        lengths[0] = L
        wmask = torch.arange(L)[None] < lengths[:, None]
        box_mask = torch.rand(B, boxes, generator=g) > 0.2
        box_mask[:, 0] = True
        emb = rnd(B, L, E, sc=0.5).requires_grad_(True)                         # what `self.embedding` emits: [BERT subword ; tag embedding]
        vis_box_feat = rnd(B, boxes, n_vis, sc=0.5).requires_grad_(True)         # the frozen Faster-RCNN region features
        token = torch.randint(0, T, (B, L), generator=g)
        tag = torch.randint(0, 7, (B, L), generator=g)
        inputs = {"token": token, "tag": tag, "vis_box_mask": box_mask, "vis_rel_mask": True}
        vp = VarPool(seq_len=lengths, mask=wmask, tag=tag, vis_available=torch.ones(B, dtype=torch.bool))
        # every input feature and parameter sits on the bfloat16 grid (the reference still computes in fp32 on them), so that a
        # bf16-storage run of the same step starts from identical numbers
        with torch.no_grad():
            all_mods = [*enc.values(), pre_match, venc, text_linear, ln, dep.head_ff, dep.child_ff, dep.root_ff, dep.dec_ff, dep.mid_ff, dep.attach_scorer,
                        dep.dec_scorer, dep.root_scorer]
            for t_ in [emb, vis_box_feat, w1, w2, b_arc, dep.token_emb, dep.root_emb, dep.dec_emb,
                       *(p_ for m in all_mods for p_ in m.parameters())]:
                t_.copy_(t_.to(torch.bfloat16).to(torch.float32))
        # ---- recorders: dropout masks, the two `_mid` tensors, the scorers' projected inputs ----
        drop_masks, mids, cap = [], [], {}

        def get_mask(x, p):
            m = (torch.rand(x.shape, generator=g) >= p).to(x.dtype) / (1 - p)
            drop_masks.append(m)
            return m
        SharedDropout.get_mask = staticmethod(get_mask)
        hooks = [pre_match.register_forward_pre_hook(lambda mod, inp: mids.append(inp[0]))]
        hooks += [m.register_forward_hook(lambda mod, inp, out, k=k: cap.__setitem__(k, out))
                  for k, m in (("x1", dep.attach_scorer.project1), ("x2", dep.attach_scorer.project2), ("y1", dep.dec_scorer.project1),
                               ("y2", dep.dec_scorer.project2))]
        # ---- JointModelBase.forward (base.py:215-241) from the frozen features ----
        p_enc = 0.33 if p_drop > 0 else 0.0                                      # encoder.dropout, vlgae.yaml:23 (the "nodrop" case runs without)
        enc_masks = []

        def enc_dropout(x):                                                      # nn.Dropout(p) in training mode, with the mask recorded
            if p_enc == 0:
                return x
            m = (torch.rand(x.shape, generator=g) >= p_enc).to(x.dtype) / (1 - p_enc)
            enc_masks.append(m)
            return x * m
        text_enc = NS(dropout=enc_dropout, shared_dropout=torch.nn.Identity(), linear=text_linear)
        vis_all = venc({"vis_box_feat": vis_box_feat}, vp)                       # base.py:229
        # (the encoder always emits `rel`, and `attr` with use_attr; `vis_feat_unprune` reads what cfg.add_rel / add_attr select)
        vis_enc = {k: t for k, t in vis_all.items() if k == "box" or k in factors}
        encoded = {f"vis_{k}": t for k, t in vis_enc.items()}                    # base.py:233-234
        encoded |= MLPEncoder.forward(text_enc, emb, vp)                         # base.py:68 (ModelBase.forward)
        encoded["emb"] = emb                                                     # base.py:69
        enc_x = encoded["x"]
        score = JB._forward(me, inputs, encoded, vp)
        assert encoded["x"] is enc_x                                             # the fuse went into a copy (replace: false)
        score = {**score, **JB._vis_forward(me, inputs, vis_enc, encoded, score, vp)}
        total, parts = JB.loss(me, score, {}, vp)
        loss = reduce_loss("token", total, vp.num_token, vp.batch_size)
        SharedDropout.get_mask = real_get_mask
        for hk in hooks:
            hk.remove()
        assert len(mids) == 2 and len(drop_masks) == (4 if p_drop > 0 else 0)
        # ---- gradients of the reduced loss ----
        ff = dict(head_ff=dep.head_ff, child_ff=dep.child_ff, root_ff=dep.root_ff, dec_ff=dep.dec_ff, mid_ff=dep.mid_ff,
                  attach_scorer=dep.attach_scorer, dec_scorer=dep.dec_scorer, root_scorer=dep.root_scorer)
        ff_params = {f"ff.{mod}.{k}": t for mod, m in ff.items() for k, t in m.named_parameters()}
        vis_params = {f"vis.{m}.{k}": t for m in ("box_fc", "rel_fc", "attr_fc") if hasattr(venc, m) for k, t in getattr(venc, m).linear.named_parameters()}
        named = {"emb": emb, "vis_box_feat": vis_box_feat, "w_text": text_linear.weight, **vis_params,
                 "w_word": enc["word"].linear.weight, "b_word": enc["word"].linear.bias, "w_child": enc["child"].linear.weight,
                 "b_child": enc["child"].linear.bias, "w_parent": enc["parent"].linear.weight, "b_parent": enc["parent"].linear.bias,
                 "w1": w1, "w2": w2, "b_arc": b_arc, "w_vis": pre_match.weight, "ln_w": ln.weight, "ln_b": ln.bias,
                 "token_emb": dep.token_emb, "root_emb": dep.root_emb, "dec_emb": dep.dec_emb, **ff_params}
        extra_t = [mids[0], mids[1], cap["x1"], cap["x2"], cap["y1"], cap["y2"], score["merged_dec"], score["merged_attach"], enc_x]
        grads = torch.autograd.grad(loss, list(named.values()) + extra_t, allow_unused=True)
        gn = dict(zip(named, grads[:len(named)]))
        g_mid0, g_mid1, g_x1, g_x2, g_y1, g_y2, g_md, g_ma, g_enc_x = grads[len(named):]
        # the fused x (the copy `_forward` handed to the parser) and the heads, recomputed with the reference's lines on the same tensors
        with torch.no_grad():
            vis0 = pre_match(mids[0])
            x_word = torch.cat([(enc_x.masked_fill(~wmask.unsqueeze(2), 0).sum(1) / lengths.unsqueeze(1)).unsqueeze(1), enc_x], 1)
            word0 = enc["word"].linear(x_word) * (drop_masks[0].unsqueeze(1) if p_drop > 0 else 1.0)
            att = torch.einsum("bvd, bqd -> bqv", vis0, word0[:, 1:]).softmax(2)
            x_fused = ln(enc_x + torch.einsum("bqv,bvh->bqh", att, mids[0]))
        arc = ts.DMV1o([score["merged_dec"].detach().requires_grad_(), score["merged_attach"].detach().requires_grad_()],
                       lengths).argmax.sum(-1).nonzero()
        predicted = torch.zeros(B, L + 1, dtype=torch.long)
        predicted[arc[:, 0], arc[:, 2]] = arc[:, 1]
        txt, tmask, tmarg = score["txt_packed"]
        vis, vmask, split = score["vis_packed"]
        out = {f"g_{k}": _np(t) for k, t in gn.items() if t is not None and k != "w1"}
        unused = [k for k, t in gn.items() if t is None]
        if rank:
            out.update(w1_u=_np(u), w1_v=_np(v), w1_z=_np(z), g_w1_sample=_np(gn["w1"][::5, ::7, ::3]))
        else:
            out.update(w1=_np(w1), g_w1=_np(gn["w1"]))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            **{k: _np(t) for k, t in named.items() if k != "w1"}, **out,
            lengths=_np(lengths), token=_np(token), tag=_np(tag), box_mask=_np(box_mask), factor_names=np.array(me.vis_factor_names),
            vis_split=np.array(split, dtype=np.int64), pos_for_obj=_np(pos["obj"]), pos_for_rel=_np(pos["rel"]), pos_for_attr=_np(pos["attr"]),
            drop_masks=(np.stack([_np(m) for m in drop_masks]) if drop_masks else np.zeros((0, B, d), np.float32)),
            p_drop=np.float32(p_drop), alpha=np.float32(me.cfg.grounding_interpolation), vis2txt_weight=np.float32(1.0),
            slope=np.float32(enc["child"].activation.negative_slope), ln_eps=np.float32(ln.eps), neg_inf=np.float32(-src.INF),
            n_bottleneck=np.int64(nb), unused=np.array(unused),
            enc_x=_np(enc_x), g_enc_x=_np(g_enc_x), p_enc=np.float32(p_enc), n_vis=np.int64(n_vis),
            enc_drop_mask=(_np(enc_masks[0]) if enc_masks else np.zeros((0, L, E), np.float32)),
            vis_mid=_np(mids[0]), g_vis_mid=_np(g_mid0 + g_mid1), vis_mask=_np(vmask), x_fused=_np(x_fused),
            sc_x1=_np(cap["x1"]), sc_x2=_np(cap["x2"][0]), sc_y1=_np(cap["y1"]), sc_y2=_np(cap["y2"][0]), root_rule=_np(score["root_rule"][0]),
            g_sc_x1=_np(g_x1), g_sc_x2=_np(g_x2[0]), g_sc_y1=_np(g_y1), g_sc_y2=_np(g_y2[0]),
            merged_dec=_np(score["merged_dec"]), merged_attach=_np(score["merged_attach"]), g_merged_dec=_np(g_md), g_merged_attach=_np(g_ma),
            predicted=_np(predicted), txt=_np(txt), txt_mask=_np(tmask), txt_marginal=_np(tmarg), vis_feat=_np(vis),
            dep_loss=_np(parts["nll"]), loss_txt2vis=_np(parts["txt2vis"]), loss_vis2txt=_np(parts["mt_vis2txt"]), total=_np(total),
            loss=_np(loss), num_token=_np(vp.num_token))
        print(f"{name}: loss {float(loss):.6f} total {float(total):.4f} nll {float(parts['nll']):.4f} unused {unused}")


def _csr(lists):
    return (np.cumsum([0] + [len(l) for l in lists]).astype(np.int64),
            np.asarray([i for l in lists for i in l], dtype=np.int64))


def feed_cases():
    """The data feed, the reference's own classes executed:
    ConstantTokenNumSampler (datamodule/sampler.py:15-191) -- buckets from its k-means and three epochs of batches under a
    fixed torch seed; _COCODetFeatLazyLoader.__call__ (datamodule/task/vlparse.py:29-114) on small synthetic .npy files
    (their contents are in the fixture; the test writes them back to disk)."""
    import pathlib
    import tempfile
    import _ref_import
    _ref_import.import_joint()
    from src.datamodule.sampler import ConstantTokenNumSampler
    from src.datamodule.task.vlparse import _COCODetFeatLazyLoader
    rng = np.random.default_rng(7)
    #       name                         n   buckets max_token max_sent single sort   shuffle same_len  length law
    for name, n, nb, mt, ms, thr, sib, shuffle, same, lam in (
            ("feed_sampler_n600_b16_s0", 600, 16, 500, -1, -1, True, True, False, 11),
            ("feed_sampler_n400_b8_thr_s1", 400, 8, 160, 6, 24, True, True, False, 14),
            ("feed_sampler_n60_b40_few_s2", 60, 40, 64, -1, -1, False, True, False, 2),     # fewer distinct lengths than buckets
            ("feed_sampler_n300_same_len_s3", 300, 4, 100, -1, -1, True, False, True, 5)):
        seed = int(name[-1])
        lens = np.clip(rng.poisson(lam, n) + 1, 1, 50).tolist()
        torch.manual_seed(seed)
        sm = ConstantTokenNumSampler(lens, mt, ms, nb, thr, sib, shuffle, same)
        epochs = [list(sm) for _ in range(3)]
        boff, bitems = _csr(sm.buckets)
        out = dict(seq_len=np.asarray(lens, np.int32), torch_seed=seed, num_bucket=nb, max_token=mt, max_sentence=ms,
                   single_sent_threshold=thr, sort_in_batch=sib, shuffle=shuffle, force_same_len=same,
                   sizes=np.asarray(sm.sizes, np.float64), bucket_offsets=boff, bucket_items=bitems,
                   chunks=np.asarray(sm.chunks, np.int64))
        for e, batches in enumerate(epochs):
            out[f"epoch{e}_offsets"], out[f"epoch{e}_items"] = _csr(batches)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: {len(sm.buckets)} buckets, {[len(b) for b in epochs]} batches")
    rows = [36, 4, 9, 1, 2]
    dts = [np.float16, np.float32, np.float64, np.float16, np.float32]
    files = [(rng.integers(-4, 5, (r, 2052)) / 2).astype(dt) for r, dt in zip(rows, dts)]   # coarse values: the fixture compresses
    sg = {i: {"obj": list(range(40)), "rel": [{"subj": int(a), "obj": int(b)} for a, b in rng.integers(0, 40, (25, 2))]}
          for i in range(len(rows))}
    sg[4] = {"obj": [], "rel": []}
    order = [3, 1, 4, 0, 2]
    with tempfile.TemporaryDirectory() as td:
        root = pathlib.Path(td)
        for i, f in enumerate(files):
            np.save(root / f"{i}.npy", f)
        batch = [(k, {"img_id": i}) for k, i in enumerate(order)]
        out = {f"file{i}": f for i, f in enumerate(files)}
        out["order"] = np.asarray(order)
        out["sg_rel"] = np.asarray([[i, r["subj"], r["obj"]] for i in sg for r in sg[i]["rel"]], np.int64)
        out["sg_nobj"] = np.asarray([len(sg[i]["obj"]) for i in range(len(rows))], np.int64)
        for tag, sample, gold in (("lead", 0, False), ("sample6", 6, False), ("gold", 0, True), ("gold_sample6", 6, True)):
            np.random.seed(5)
            a, b = _COCODetFeatLazyLoader(root, sg, sample, gold)(batch)
            if not gold:
                out[f"{tag}_feat_f16"] = _np(a["vis_box_feat"]).astype(np.float16)   # the inputs are f16-representable: lossless
                assert (out[f"{tag}_feat_f16"].astype(np.float32) == _np(a["vis_box_feat"])).all()
            else:   # the gold graph changes the masks only: same features as the plain call with the same sample
                assert (_np(a["vis_box_feat"]) == out[("lead" if sample == 0 else "sample6") + "_feat_f16"]).all()
            out[f"{tag}_mask"], out[f"{tag}_rel"] = _np(a["vis_box_mask"]), _np(a["vis_rel_mask"])
            out[f"{tag}_available"], out[f"{tag}_box"] = _np(a["vis_available"]), _np(b["vis_box"])
        np.savez_compressed(os.path.join(HERE, "feed_collate_5img_s5.npz"), **out)
        print("feed_collate_5img_s5:", {k: v.shape for k, v in out.items() if k.endswith("feat_f16")})


if __name__ == "__main__":
    if len(sys.argv) > 1:   # `make_golden.py langfeat_cases [...]`: only the named generator functions
        for fn_name in sys.argv[1:]:
            globals()[fn_name]()
        sys.exit(0)
    dmv_case("dmv_B4_L10_s0", 0, 4, 10, "rand", store_merged=True)
    dmv_case("dmv_B4_L10_s1_full", 1, 4, 10, "full")
    dmv_case("dmv_B5_L7_s2", 2, 5, 7, [7, 5, 3, 1, 6], normalise_attach=True)
    dmv_case("dmv_B3_L1_s3", 3, 3, 1, "full")
    dmv_case("dmv_B3_L2_s4", 4, 3, 2, [2, 1, 2])
    dmv_case("dmv_B8_L40_s0", 0, 8, 40, "rand")
    dmv_case("dmv_B4_L40_s1_full", 1, 4, 40, "full", normalise_attach=True)
    dmv_case("dmv_B4_L80_s0", 0, 4, 80, "rand")
    rules_case("rules_B4_L10_T7_s0", 0, 4, 10, 7, "rand", use_mask=True)
    rules_case("rules_B3_L8_T3_s1_rootps", 1, 3, 8, 3, "rand", use_mask=False, root_per_sentence=True)
    rules_case("rules_B4_L40_T45_s2", 2, 4, 40, 45, "rand", use_mask=True)
    deptree_case("deptree_B4_N6_s0_enum", 0, 4, 6, "full", enumerate_=True)
    deptree_case("deptree_B3_N5_s1_enum", 1, 3, 5, "full", enumerate_=True, scale=3.0)
    deptree_case("deptree_B4_N11_s0", 0, 4, 11, "rand")
    deptree_case("deptree_B8_N41_s1", 1, 8, 41, "rand")
    deptree_case("deptree_B2_N81_s2", 2, 2, 81, "rand")
    deptree_case("deptree_B3_N2_s3", 3, 3, 2, "full")
    align_cases()
    attnfuse_cases()
    grounding_cases()
    decode_cases()
    reduced_cases()
    arcenc_cases()
    boxrel_cases()
    langfeat_cases()
    scorer_cases()
    trainstep_cases()
    feed_cases()
