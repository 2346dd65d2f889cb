"""One VLGAE training step (BASELINE.json configs[4]: "frozen BERT + Faster-RCNN feats -> biaffine -> inside-outside marginal loss")
from the FROZEN FEATURES to the gradient of every trainable parameter, wired AS THE REFERENCE WIRES IT -- host-side mirror of
`JointModelBase.forward` + `DependencyBoxRel.loss` + `reduce_loss` + `loss.backward()`.  Paths are relative to /root/reference.

Contract.  `build(...)` returns `step()`; `step()` runs forward + backward of the chain below on the current HIP stream (no host
synchronisation, no allocation through the driver: capturable as one HIP graph) and returns (reduced loss, {leaf name: gradient}, ()).
The SAME function is what
  * `tests/test_gpu_parity.py::test_training_step_reference_wiring[*]` runs on fixtures produced by the reference's own methods
    (tests/golden/trainstep_*.npz, make_golden.py `trainstep_cases`), in float32 and in the bf16 configuration the bench times,
  * `..._config_size[*]` runs at B = 256 against the reference's formulation in float64 torch ops,
  * `bench.py`'s `train_step` entries time (vlgae_amd/bench/secondary.py), and
  * `bench.py --workload train_step [--gpus N]` shards (vlgae_amd/bench/sharded_step.py).

  JointModelBase.forward           src/model/base.py:215-241
    VisBoxRelSimpleEncoder.forward src/model/vis_encoder/box_rel.py:29-52   base.py:229   box_fc / rel_fc / attr_fc on [box ; mean box],
      + the cat of vis_feat_unprune src/model/joint.py:143-171              one [B,V,H] factor tensor (obj | rel | attr | img)        -> encoders.vis_box_rel_encoder
    MLPEncoder.forward             src/model/text_encoder/mlp_encoder.py:36-40   base.py:68   x = Linear(Dropout(emb))               -> encoders.mlp_encoder
    DependencyBoxRel._forward      src/model/joint.py:658-675
      vis_feat_unprune             :137-178   vis_feat = vis_mlp_pre_matching(vis_mid)  (bias-free nn.Linear)          -> align.linear
      lang_feat_word_only          :193-211   root row = masked mean, word encoder (+ SharedDropout)                  -> langfeat.lang_feat_word_only
      attention fuse               :670-674   softmax(vis_feat . word^T) vis_mid, residual, LayerNorm                  -> align.attention_fuse
        the fused x goes into a COPY of `encoded` (feat_fuse_args.replace: false, :376-377): only the parser sees it
    DiscriminativeNDMV._forward    src/model/ldndmv.py:171-216  on that copy
      context_mode 'mean'          :226,:254  h = cat([emb, mean_l(fused x)])
      head_ff / child_ff / root_ff / dec_ff (MLP, nn/common.py:23-51), mid_ff (DMVSkipConnectEncoder, nn/dmv_spec.py:6-54),
      the scorers' project1 / project2 (nn.Linear, nn/dmv_spec.py:63-68)                                               -> parser_ff.parser_feed_forward
                                                                     (module by module in torch: `scorer_feed_forward` below, the tests' comparison)
      scores -> log-softmax over tokens -> gather / direction select / root gather / merge   :184-209                   -> scorer.ndmv_potentials
    DependencyBoxRel._vis_forward  joint.py:677-691
      lang_feat_max_tree           :235-292   on the caller's `encoded`, i.e. the UN-fused x: DMV1o marginals + Viterbi heads of
                                              the detached potentials, word | child | parent encoders (+ SharedDropout), arc encoder -> langfeat.lang_feat_max_tree
      gather_logit_simple          :406-419   } never materialised: alignment maxima + arg-max on the matrix cores,
  DependencyBoxRel.loss            :693-711   }
      loss_grounding_factor_ce     :439-491   } POS prior (use_pos_prior: true), both cross-entropies, vis2txt = 1          -> align.grounding_loss_factor_ce
      DiscriminativeNDMV.loss      ldndmv.py:277-281  viterbi_training: -DMV1o(potentials).max.sum() (the step's one Viterbi pass is reused)
      alpha * mt_loss + (1 - alpha) * dep_loss, alpha = grounding_interpolation = 0.5 (config/model/vlgae.yaml:67)
  reduce_loss('token')             src/utility/fn.py:50-56 (src/pipeline.py:124,249-250): / (num_token + 1e-12)
  loss.backward()                  every adjoint of the above

Leaves: the frozen features `emb` [B,L,E] (= what `self.embedding` emits: [BERT subword ; tag embedding]; its gradient is computed, the
tag embedding is trainable) and `vis_box_feat` [B,R,n] (no gradient unless `feature_grads`), and every trainable tensor: `w_text`
(MLPEncoder.linear), `w_venc` / `b_venc` (the visual encoder's box_fc | rel_fc | attr_fc stacked), `w_vis`, `w_enc` / `b_enc` (word | child |
parent encoders stacked), `ln_w` / `ln_b`, `w1` / `w2` / `b` (arc encoder), `token_emb` / `root_emb` / `dec_emb` and the "ff.*" feed-forwards.

`wiring="r3"` keeps round 3's chain for continuity of the bench history (fused x fed to lang_feat_max_tree, scorer inputs and
matching-space features as leaves, plain sum of the two losses): NOT what the reference does; see HISTORY.md section 5.
"""
import torch
import torch.nn.functional as F

SLOPE = 0.01           # nn.LeakyReLU() default, nn/common.py:31
FF_MODULES = ("head_ff", "child_ff", "root_ff", "dec_ff", "mid_ff", "attach_scorer", "dec_scorer", "root_scorer")


# ----------------------------------------------------------------------------------------------------------------------------
# The parser's feed-forwards (out of the hot path: plain Linear / LeakyReLU stacks, left to the library).  Parameter names are
# the reference modules' own `named_parameters()` names behind "ff.<module>." so that a fixture's tensors drop in.
# ----------------------------------------------------------------------------------------------------------------------------
def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _mlp(P, name, x, drop=None):
    """MLP (nn/common.py:23-51): Linear -> LeakyReLU -> SharedDropout (a mask broadcastable to the output, or None)."""
    y = F.leaky_relu(_lin(P, f"ff.{name}.linear", x), SLOPE)
    return y if drop is None else y * drop.to(y.dtype)


def _bottleneck(P, name, x):
    if name + ".weight" in P:                                   # n_bottleneck == 0: one Linear
        return _lin(P, name, x)
    return _lin(P, name + ".1", _lin(P, name + ".0", x))      # nn.Sequential(Linear(H, nb), Linear(nb, H)), no activation between


def _skip_connect(P, x, drop=None):
    """DMVSkipConnectEncoder.forward (nn/dmv_spec.py:38-54): [..., H] -> [..., dir, val, H].  drop: nn.Dropout's mask [..., 2, 2, H] or None."""
    act = lambda t: F.leaky_relu(t, SLOPE)
    m = "ff.mid_ff."
    has_child = _bottleneck(P, m + "HASCHILD_linear", x) + x
    no_child = _bottleneck(P, m + "NOCHILD_linear", x) + x
    h = torch.stack([no_child, has_child], dim=-2)
    h = act(_lin(P, m + "valence_linear", act(h)))
    x4 = x.unsqueeze(-2)
    left = _bottleneck(P, m + "LEFT_linear", h) + x4
    right = _bottleneck(P, m + "RIGHT_linear", h) + x4
    h = torch.stack([left, right], dim=-3)
    h = act(_lin(P, m + "direction_linear", act(h)))
    if drop is not None:
        h = h * drop.to(h.dtype)
    return _lin(P, m + "linear2", act(_lin(P, m + "linear1", h)))


def scorer_feed_forward(P, emb, x_fused, drop_head=None, drop_small=None, drop_mid=None):
    """ldndmv.py:174-205 up to the scorers' projected inputs, MODULE BY MODULE as the reference runs it: (x1 [B,L,2,2,r], x2 [T,2,2,r],
    y1 [B,L,2,2,r], y2 [2,2,2,r], root_rule [T]).  context_mode 'mean' (:226): every token's representation is cat([emb, mean over ALL L
    positions of x]).  Dropout masks as in vlgae_amd.parser_ff.parser_feed_forward (explicit, so that the two formulations can be compared)."""
    B, L, _ = emb.shape
    T = P["token_emb"].shape[0]
    H = P["ff.head_ff.linear.weight"].shape[0]
    M0 = B * L
    ctx = x_fused.mean(1, keepdim=True).expand(-1, L, -1)
    h = torch.cat([emb, ctx.to(emb.dtype)], dim=-1)
    ds = (lambda a, b: None) if drop_small is None else (lambda a, b: drop_small[a:b].unsqueeze(1))
    dm = (lambda a, b, shp: None) if drop_mid is None else (lambda a, b, shp: drop_mid[4 * a:4 * b].reshape(*shp, 2, 2, H))
    h_parent = _skip_connect(P, _mlp(P, "head_ff", h, None if drop_head is None else drop_head.unsqueeze(1)), dm(0, M0, (B, L)))
    h_child = _skip_connect(P, _mlp(P, "child_ff", P["token_emb"], ds(0, T)), dm(M0, M0 + T, (T,)))                  # [T,2,2,H]
    h_root = _skip_connect(P, _mlp(P, "root_ff", P["root_emb"], ds(T, T + 1)), dm(M0 + T, M0 + T + 1, (1,)))        # [1,2,2,H]
    h_dec = _skip_connect(P, _mlp(P, "dec_ff", P["dec_emb"], ds(T + 1, T + 3)), dm(M0 + T + 1, M0 + T + 3, (2,)))   # [2,2,2,H]
    x1, x2 = _lin(P, "ff.attach_scorer.project1", h_parent), _lin(P, "ff.attach_scorer.project2", h_child)
    y1, y2 = _lin(P, "ff.dec_scorer.project1", h_parent), _lin(P, "ff.dec_scorer.project2", h_dec)
    r1, r2 = _lin(P, "ff.root_scorer.project1", h_root), _lin(P, "ff.root_scorer.project2", h_child)
    root_rule = torch.einsum("hdve,cdve->hc", r1.float(), r2.float()).log_softmax(-1)[0]   # :205: sum over (dir, val), softmax over tokens
    return x1, x2, y1, y2, root_rule


def init_feed_forward(g, dev, dtype, E, h, Et, T, H, nb, r):
    """Random parameters with the reference modules' shapes (vlgae.yaml: H = 256, n_bottleneck = 150, ranks 16)."""
    P = {}

    def lin(name, n_in, n_out, bias=True):
        P[name + ".weight"] = (torch.randn(n_out, n_in, generator=g) * n_in ** -0.5).to(dev, dtype).requires_grad_(True)
        if bias:
            P[name + ".bias"] = (torch.randn(n_out, generator=g) * 0.1).to(dev, dtype).requires_grad_(True)

    for name, n_in in (("head_ff", E + h), ("child_ff", Et), ("root_ff", 10), ("dec_ff", 10)):
        lin(f"ff.{name}.linear", n_in, H)
    for name in ("HASCHILD_linear", "NOCHILD_linear", "LEFT_linear", "RIGHT_linear"):
        if nb:
            lin(f"ff.mid_ff.{name}.0", H, nb)
            lin(f"ff.mid_ff.{name}.1", nb, H)
        else:
            lin(f"ff.mid_ff.{name}", H, H)
    for name in ("valence_linear", "direction_linear", "linear1", "linear2"):
        lin(f"ff.mid_ff.{name}", H, H)
    for name in ("attach_scorer", "dec_scorer", "root_scorer"):
        lin(f"ff.{name}.project1", H, r)
        lin(f"ff.{name}.project2", H, r)
    P["token_emb"] = torch.randn(T, Et, generator=g).to(dev, dtype).requires_grad_(True)
    P["root_emb"] = torch.randn(1, 10, generator=g).to(dev, dtype).requires_grad_(True)
    P["dec_emb"] = torch.randn(2, 10, generator=g).to(dev, dtype).requires_grad_(True)
    return P


def forced_tree_score(md, ma, heads, lengths, big=1e4):
    """Score of the dependency tree `heads` [B,N] (heads[b,c] = head of word c, 0 = the root token) under root-merged DMV1o potentials
    md [B,N,2,2,2] / ma [B,N,N,2 (head, child, valence)]: [B,1], differentiable w.r.t. both (its gradient is the tree's derivation
    counts).  The counts come from the Max-semiring DP itself on potentials whose arcs (heads[c] -> c) carry a bonus that no other tree
    can make up: the valence of every attachment and every continue / stop decision are functions of the tree, so the arg-max derivation
    of the biased potentials IS the derivation of that tree, under exactly the conventions the kernel uses."""
    import vlgae_amd.torch_struct as ts
    B, N = heads.shape
    with torch.no_grad():
        child = torch.arange(N, device=heads.device)[None].expand(B, -1)
        live = (child >= 1) & (child <= lengths[:, None])
        bonus = torch.zeros((B, N, N), dtype=torch.float32, device=heads.device)
        b_idx = torch.arange(B, device=heads.device)[:, None].expand(-1, N)
        bonus[b_idx[live], heads[live], child[live]] = big
    with torch.enable_grad():
        pot = [md.detach().float().requires_grad_(True), (ma.detach().float() + bonus.unsqueeze(-1)).requires_grad_(True)]
        cd, ca = torch.autograd.grad(ts.DMV1o(pot, lengths).max.sum(), pot)
    ct = torch.float64 if md.dtype == torch.float64 else torch.float32   # (the tests' float64 formulation keeps its precision)
    return ((cd.to(ct) * md.to(ct)).flatten(1).sum(1) + (ca.to(ct) * ma.to(ct)).flatten(1).sum(1)).view(B, 1)


# ----------------------------------------------------------------------------------------------------------------------------
def build(B, L, R, dev, dtype=torch.bfloat16, d=128, h=256, seed=11, with_scorer=True, T=45, r=16, wiring="reference", given=None,
          alpha=0.5, use_pos_prior=True, vis2txt=1.0, p_drop=0.33, E=800, Et=32, H=256, nb=150, p_ff_drop=0.33, p_mid_drop=0.3,
          factors=(), n_vis=2048, p_enc=0.33, pos_for=None, ln_eps=1e-5, ff_dtype=None, fused_ff=True, feature_grads=False):
    """The step function of one training step at B sentences of <= L words and R region boxes per image.

    factors: which of ("rel", "attr", "img") the model adds to the object factor (cfg.add_rel / add_attr / add_image; the shipped
    config/model/vlgae.yaml:39-41 has all three: V = R + R^2 + R + 1 columns, 1369 at R = 36; BASELINE.json configs[1] names R = 36
    region columns: the object factor alone, the default).  n_vis / E: widths of the frozen region features / embeddings (2048 / 800).
    `given` (a dict) replaces any of the synthetic inputs / parameters by name (tests: the fixture's tensors) -- features emb [B,L,E],
    vis_box_feat [B,R,n_vis]; batch lengths / token / tag [B,L], box_mask [B,R], drop [4,B,d] (the SharedDropout masks in the reference's
    call order: word-only, then word | child | parent; or None), enc_drop [B,L,E] (MLPEncoder's nn.Dropout mask; or None), heads [B,L+1] (a tree
    to use INSTEAD of the Viterbi tree of the step's own potentials -- teacher forcing, see `step.forced_heads` below); parameters
    w_text [h,E], w_venc [F h, 2 n_vis] / b_venc [F h], w_vis [d,h], w_enc [3d,h], b_enc [3d], ln_w, ln_b [h], w1 [d,d,d], w2 [d,d], b [d],
    token_emb / root_emb / dec_emb and the "ff.*" feed-forward parameters.  With `given` drop / enc_drop absent, fresh masks are drawn
    every step (p_drop / p_enc; the embedding dropout from a device-resident counter-based generator, encoders.DeviceRng).
    ff_dtype: storage / compute type of the parser's feed-forwards and of `emb` (default = dtype); fused_ff: vlgae_amd.parser_ff (folded /
    fused library GEMMs, hand-written adjoint) instead of the module-by-module torch formulation `scorer_feed_forward` (same values).
    p_ff_drop / p_mid_drop: the parser feed-forwards' dropout (shipped: 0.33 / 0.3; masks drawn per step; 0 = off, as in the fixtures).
    alpha / use_pos_prior / vis2txt: config/model/vlgae.yaml:62-67.  feature_grads: also return the gradient w.r.t. vis_box_feat (the
    reference does not compute it -- the region features are data; the parity tests do).
    Returns step(); step() -> (loss, {name: gradient}, ()).

    wiring="r3": round 3's chain (see the module docstring); `with_scorer` only matters there, and R is its V."""
    if wiring == "r3":
        return _build_r3(B, L, R, dev, dtype, d, h, seed, with_scorer, T, r)
    if wiring != "reference":
        raise ValueError(wiring)
    import vlgae_amd.torch_struct as ts
    from vlgae_amd import align, encoders, langfeat, parser_ff, scorer
    N, Q = L + 1, 2 * (L + 1)
    given = dict(given or {})
    factors = tuple(factors)
    if any(f not in ("rel", "attr", "img") for f in factors):
        raise ValueError(f"train_step.build: factors {factors}")
    add_rel, add_attr, add_image = "rel" in factors, "attr" in factors, "img" in factors
    _, V, vis_split, factor_names = encoders.factor_layout(R, add_rel, add_attr, add_image)
    n_enc = 1 + add_rel + add_attr
    ff_dtype = dtype if ff_dtype is None else ff_dtype
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc

    def leaf(name, make, dt=dtype, grad=True):
        t = given.pop(name) if name in given else make()
        return t.detach().to(dev, dt).contiguous().requires_grad_(grad)

    # ---- the frozen features (BERT subword + tag embedding; Faster-RCNN region features) and every trainable weight behind them ----
    P = dict(
        emb=leaf("emb", lambda: rnd(B, L, E, sc=0.5), ff_dtype), vis_box_feat=leaf("vis_box_feat", lambda: rnd(B, R, n_vis, sc=0.5), grad=feature_grads),
        w_text=leaf("w_text", lambda: rnd(h, E, sc=E ** -0.5), ff_dtype),
        w_venc=leaf("w_venc", lambda: rnd(n_enc * h, 2 * n_vis, sc=(2 * n_vis) ** -0.5)), b_venc=leaf("b_venc", lambda: rnd(n_enc * h, sc=0.1)),
        w_vis=leaf("w_vis", lambda: rnd(d, h, sc=h ** -0.5)),
        w_enc=leaf("w_enc", lambda: rnd(3 * d, h, sc=h ** -0.5)), b_enc=leaf("b_enc", lambda: rnd(3 * d, sc=0.1)),
        ln_w=leaf("ln_w", lambda: torch.ones(h), torch.float32), ln_b=leaf("ln_b", lambda: torch.zeros(h), torch.float32),
        w1=leaf("w1", lambda: rnd(d, d, d, sc=1.0 / d)), w2=leaf("w2", lambda: rnd(d, d, sc=d ** -0.5)), b=leaf("b", lambda: rnd(d, sc=0.1)),
    )
    ff_given = {k: given.pop(k) for k in list(given) if k.startswith("ff.") or k in ("token_emb", "root_emb", "dec_emb")}
    if ff_given:
        P.update({k: t.detach().to(dev, ff_dtype).contiguous().requires_grad_(True) for k, t in ff_given.items()})
    else:
        P.update(init_feed_forward(g, dev, ff_dtype, E, h, Et, T, H, nb, r))
    # ---- the batch ----
    if "lengths" in given:
        lengths = given.pop("lengths").to(dev, torch.int64)
    else:
        lengths = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
        lengths[0] = L
        lengths = lengths.to(dev)
    token = given.pop("token").to(dev) if "token" in given else torch.randint(0, P["token_emb"].shape[0], (B, L), generator=g).to(dev)
    tag = given.pop("tag").to(dev) if "tag" in given else torch.randint(0, 7, (B, L), generator=g).to(dev)
    if "box_mask" in given:
        box_mask = given.pop("box_mask").to(dev, torch.bool)
    else:   # ragged region lists as the reference's collate builds them: image i has n_i <= R boxes, `masks_output[i, :n_i] = True` and
        # padding behind them (src/datamodule/task/vlparse.py:68-83) -- a PREFIX mask per image, n_i drawn from [0.6 R, R]
        n_box = torch.randint(max(1, (3 * R) // 5), R + 1, (B,), generator=g)
        box_mask = (torch.arange(R)[None] < n_box[:, None]).to(dev)
    vmask = encoders.factor_mask(box_mask, add_rel, add_attr, add_image)       # vis_feat_unprune's mask (joint.py:140-170): data, built once per batch
    fixed_drop = given.pop("drop") if "drop" in given else "draw"
    if fixed_drop is not None and not isinstance(fixed_drop, str):
        fixed_drop = fixed_drop.to(dev, torch.float32).permute(1, 0, 2).contiguous()      # [B,4,d]
    forced_heads = given.pop("heads").to(dev, torch.int64) if "heads" in given else None
    enc_drop = given.pop("enc_drop") if "enc_drop" in given else "draw"
    if enc_drop is not None and not isinstance(enc_drop, str):
        enc_drop = enc_drop.to(dev, torch.float32).contiguous()                           # [B,L,E]
    if given:
        raise ValueError(f"train_step.build: unknown given entries {sorted(given)}")
    rng = encoders.DeviceRng(seed * 7919 + 17, dev)       # the counter-based dropout draws of the step (advanced on the device once per step)
    if pos_for is None:
        pos_for = dict(obj=torch.tensor([0, 1, 2]), rel=torch.tensor([2, 3]), attr=torch.tensor([4]))
    pos_for = {k: t.to(dev) for k, t in pos_for.items()}
    num_token = lengths.sum()                                     # a 0-d tensor like vp.num_token (var_pool.py:18)
    num_token_f = float(num_token.item())
    names = sorted(k for k in P if P[k].requires_grad)
    leaves = [P[k] for k in names]
    aux = {}
    # the POS prior table (joint.py:446-470) is a function of the batch's tags only -- data, like the masks: built once per batch
    pen = seg = None
    if use_pos_prior:
        pen, seg = align.grounding_prior(tag, factor_names, vis_split, pos_for, Q)
    # loss = (alpha mt + (1 - alpha) dep) / (num_token + 1e-12), dep = -sum_b max_b: two per-batch coefficients.  They seed the
    # backward pass directly (autograd.grad's grad_outputs), so the scalar arithmetic of the combination has no adjoint launches.
    coef = (torch.tensor([alpha, -(1.0 - alpha)], dtype=torch.float32, device=dev) / (num_token.to(torch.float32) + 1e-12))
    c_mt, c_max = coef[0], coef[1]
    seed_max = c_max.repeat(B)
    T_, H_ = P["token_emb"].shape[0], P["ff.head_ff.linear.weight"].shape[0]

    def draw_masks():
        """(drop [B,4,d] or None, parser_ff masks): the per-sentence SharedDropout masks of one step come out of ONE launch of the
        counter-based generator when the rates agree; mid_ff's nn.Dropout is drawn inside its activation kernel (no tensor)."""
        mid = dict(mid_rng=rng, p_mid=p_mid_drop) if p_mid_drop > 0 else {}
        if isinstance(fixed_drop, str) and p_drop == p_ff_drop and 0 < p_drop < 1:
            n0, n1 = B * 4 * d, B * H_
            buf = encoders.dropout_mask(rng, encoders.SITE_SHARED, p_drop, n0 + n1 + T_ + 3)
            return buf[:n0].view(B, 4, d), dict(drop_head=buf[n0:n0 + n1].view(B, H_), drop_small=buf[n0 + n1:], **mid)
        if isinstance(fixed_drop, str):
            drop = encoders.dropout_mask(rng, encoders.SITE_SHARED, p_drop, B * 4 * d).view(B, 4, d) if 0 < p_drop < 1 else None
        else:
            drop = fixed_drop
        ff = {}
        if 0 < p_ff_drop < 1:
            buf = encoders.dropout_mask(rng, encoders.SITE_SHARED_FF, p_ff_drop, B * H_ + T_ + 3)
            ff = dict(drop_head=buf[:B * H_].view(B, H_), drop_small=buf[B * H_:])
        return drop, dict(**ff, **mid)

    def step(stage_hook=None):
        """forward + backward; returns (reduced loss, gradients by leaf name, ()).  stage_hook (optional) is called from inside the
        backward pass once the adjoints of the DP and of the grounding loss have run (the cotangent of `txt` exists) -- where a
        data-parallel trainer starts reducing its first gradient bucket."""
        drop, ff_masks = draw_masks()
        d0, d3 = (None, None) if drop is None else (drop[:, 0:1], drop[:, 1:4])
        # ---- JointModelBase.forward, base.py:229 / :68: the two trainable encoders on the frozen features ----
        vis_mid, _, _ = encoders.vis_box_rel_encoder(P["vis_box_feat"], P["w_venc"], P["b_venc"], add_rel, add_attr, add_image, SLOPE)
        if enc_drop is None or p_enc == 0:
            enc_x = encoders.mlp_encoder(P["emb"], P["w_text"], training=False)
        elif isinstance(enc_drop, str):
            enc_x = encoders.mlp_encoder(P["emb"], P["w_text"], p_enc, rng=rng)
        else:
            enc_x = encoders.mlp_encoder(P["emb"], P["w_text"], p_enc, mask=enc_drop)
        if enc_x.dtype != dtype:                                          # (ff_dtype != dtype: the language side runs in `dtype`)
            enc_x = enc_x.to(dtype)
        # ---- DependencyBoxRel._forward, joint.py:658-675 ----
        vis_feat = align.linear(vis_mid, P["w_vis"])                                                         # :175 (and again :688: same values)
        # the word | child | parent encoders' Linear on cat([masked mean, x]) ONCE: joint.py:204-209 (word-only) and :262-273 (max-tree) read the
        # same un-fused encodings through the same word encoder, under two SharedDropout masks
        pre = langfeat.encoder_projection(enc_x, lengths, P["w_enc"], P["b_enc"])
        word0, _, _ = langfeat.lang_feat_word_only(None, lengths, drop=d0, pre=pre, masks=False)             # :667 (the fuse reads the features only)
        x_f = align.attention_fuse(vis_feat, word0, vis_mid, enc_x, P["ln_w"], P["ln_b"], ln_eps)             # :670-674
        # ---- DiscriminativeNDMV._forward on the fused copy, ldndmv.py:171-216 ----
        if fused_ff:   # the same mathematics with folded / fused GEMMs and a hand-written adjoint
            x1, x2, y1, y2, root_rule = parser_ff.parser_feed_forward(P, P["emb"], x_f, **ff_masks)
        else:          # module by module, as the reference runs it (explicit masks: the comparison form of the tests)
            mid = None if p_mid_drop <= 0 else encoders.dropout(torch.ones(4 * (B * L + T_ + 3), H_, device=dev), p_mid_drop, rng=rng, site=encoders.SITE_MID_FF)
            x1, x2, y1, y2, root_rule = scorer_feed_forward(P, P["emb"], x_f, ff_masks.get("drop_head"), ff_masks.get("drop_small"), mid)
        md, ma = scorer.ndmv_potentials(x1, x2, y1, y2, root_rule, token)
        # ---- DependencyBoxRel._vis_forward, joint.py:677-691: the UN-fused x; the potentials are constants of this stage (:252-253) ----
        txt, tmask, tmarg = langfeat.lang_feat_max_tree(None, lengths, md.detach(), ma.detach(), None, None, P["w1"],
                                                        P["w2"], P["b"], keep_viterbi=True, drop=d3, aux=aux, pre=pre, heads=step.forced_heads)
        if stage_hook is not None:
            txt.register_hook(lambda g_: stage_hook())
        # ---- DependencyBoxRel.loss, joint.py:693-711 ----
        mt, sums = align.grounding_loss_factor_ce(txt, vis_feat, tmask, vmask, tmarg, num_token_f, vis2txt, pen, seg)
        if step.forced_heads is None:
            mx = ts.DMV1o([md, ma], lengths).max                  # ldndmv.py:277-281: dep = -max.sum(); lang_feat_max_tree's Viterbi pass is reused
        else:
            mx = forced_tree_score(md, ma, step.forced_heads, lengths)   # teacher forcing: the given tree's score in place of the best tree's
        with torch.no_grad():                                     # alpha mt + (1 - alpha) dep, reduce_loss('token'): c_mt mt + sum_b c_max max_b, two launches
            loss = torch.addcmul(torch.dot(mx.view(-1), seed_max), c_mt, mt)
        grads = torch.autograd.grad([mt, mx], leaves, [c_mt, seed_max.view(mx.shape)])
        rng.advance()                                                 # the next step (or graph replay) draws new dropout masks at every site
        # intermediates for the parity tests, DETACHED: a reference to a previous step's autograd graph kept alive across a HIP-graph
        # capture makes torch 2.10 / ROCm 7 crash in capture_end
        step.last = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in dict(
            enc_x=enc_x, vis_mid=vis_mid, x_fused=x_f, merged_dec=md, merged_attach=ma, txt=txt, txt_mask=tmask, txt_marginal=tmarg, vis_feat=vis_feat, sums=sums,
            viterbi_max=mx, mt_loss=mt, heads=aux.get("heads")).items()}
        return loss, dict(zip(names, grads)), ()

    step.names, step.P, step.lengths, step.wiring = names, P, lengths, wiring
    # Teacher forcing (parity tests only; None = the reference's behaviour): with a tree given, lang_feat_max_tree reads ITS parents and
    # marginals, and the parser's loss is -score(that tree) instead of -max -- the same function of the parameters the reference
    # differentiates when its own Viterbi tree is that tree (joint.py:256-273 and ldndmv.py:277-281 treat the tree as a constant).  A bf16
    # run whose near-tied attachments flip is thereby compared on the reference's tree: rounding is separated from tree flips.
    step.forced_heads = forced_heads
    step.batch = dict(token=token, tag=tag, box_mask=box_mask, vis_mask=vmask, alpha=alpha, factor_names=factor_names, vis_split=vis_split,
                      factors=factors, pos_for=pos_for, use_pos_prior=use_pos_prior, vis2txt=vis2txt, enc_drop=enc_drop, p_enc=p_enc)
    step.shape = dict(B=B, L=L, R=R, V=V, d=d, h=h, E=E, n_vis=n_vis)
    step.trainable = tuple(k for k in names if k not in ("emb", "vis_box_feat"))
    # Parameters in the order their gradients become FINAL during the backward pass (autograd runs the later-created node first:
    # -max, grounding loss, lang_feat_max_tree | score construction, the parser's feed-forwards | attention fuse, word-only encoder,
    # vis_mlp_pre_matching | the text and visual encoders): what a data-parallel trainer's buckets follow.  `step.on_grad(name, grad)`, if set, is called from
    # inside the backward pass the moment a parameter's (accumulated) gradient exists.
    ff_names = [k for k in names if k.startswith("ff.") or k in ("token_emb", "root_emb", "dec_emb")]
    step.ready_groups = (["w1", "w2", "b"], ff_names, ["ln_w", "ln_b", "w_enc", "b_enc", "w_vis"], ["w_text", "w_venc", "b_venc"])
    step.on_grad = None
    for k in step.trainable:
        P[k].register_hook(lambda g_, k=k: step.on_grad(k, g_) if step.on_grad is not None else None)
    return step


# ----------------------------------------------------------------------------------------------------------------------------
def _build_r3(B, L, V, dev, dtype, d, h, seed, with_scorer, T, r):
    """Round 3's chain, unchanged (bench continuity only).  Differences from the reference's wiring: the fused x feeds
    lang_feat_max_tree, the scorers' projected inputs / vis_feat / the fuse's word features are leaves, vis_mask is all-true, no POS
    prior, no dropout, total = grounding + dep (no alpha, no token reduction)."""
    import vlgae_amd.torch_struct as ts
    from vlgae_amd import align, langfeat, scorer
    N = L + 1
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    leaf = lambda *s, sc=1.0, dt=dtype: rnd(*s, sc=sc).to(dt).requires_grad_(True)
    P = dict(
        vis_feat=leaf(B, V, d), txt_word=leaf(B, N, d), vis_mid=leaf(B, V, h), enc_x=leaf(B, L, h),
        ln_w=torch.ones(h, device=dev, requires_grad=True), ln_b=torch.zeros(h, device=dev, requires_grad=True),
        w_enc=leaf(3 * d, h, sc=h ** -0.5), b_enc=leaf(3 * d, sc=0.1),
        w1=leaf(d, d, d, sc=1.0 / d), w2=leaf(d, d, sc=d ** -0.5), b=leaf(d, sc=0.1),
    )
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
    attach = torch.randn(B, L, L, 2, generator=g).to(dev)
    root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
    md, ma = ts.DMV1o.merge(dec, attach, root)
    md, ma = md.to(dtype).contiguous(), ma.to(dtype).contiguous()
    lengths = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lengths[0] = L
    lengths = lengths.to(dev)
    vmask = torch.ones(B, V, dtype=torch.bool, device=dev)
    num_token = float(lengths.sum().item())
    if with_scorer:
        P.update(sc_x1=leaf(B, L, 2, 2, r, sc=0.5, dt=torch.float32), sc_x2=leaf(T, 2, 2, r, sc=0.5, dt=torch.float32),
                 sc_y1=leaf(B, L, 2, 2, r, sc=0.5, dt=torch.float32), sc_y2=leaf(2, 2, 2, r, sc=0.5, dt=torch.float32),
                 sc_root=torch.randn(T, generator=g).log_softmax(-1).to(dev).requires_grad_(True))
        token = torch.randint(0, T, (B, L), generator=g).to(dev)
    names = sorted(P)
    leaves = [P[k] for k in names]
    pot = [md.detach().requires_grad_(True), ma.detach().requires_grad_(True)]
    one, minus_one = torch.ones((), device=dev), torch.full((B,), -1.0, device=dev)

    def step(stage_hook=None):
        x = align.attention_fuse(P["vis_feat"], P["txt_word"], P["vis_mid"], P["enc_x"], P["ln_w"], P["ln_b"], 1e-5)
        if with_scorer:
            smd, sma = scorer.ndmv_potentials(P["sc_x1"], P["sc_x2"], P["sc_y1"], P["sc_y2"], P["sc_root"], token)
            cmd, cma, loss_pot = smd.detach(), sma.detach(), [smd, sma]
        else:
            cmd, cma, loss_pot = md, ma, pot
        txt, tmask, txt_marginal = langfeat.lang_feat_max_tree(x, lengths, cmd, cma, P["w_enc"], P["b_enc"], P["w1"], P["w2"], P["b"],
                                                               keep_viterbi=True, compute_dtype=dtype)   # (x is the fuse's fp32 output)
        if stage_hook is not None:
            txt.register_hook(lambda g_: stage_hook())
        mt, _ = align.grounding_loss_factor_ce(txt, P["vis_feat"], tmask, vmask, txt_marginal, num_token, 1.0)
        mx = ts.DMV1o(loss_pot, lengths).max
        with torch.no_grad():
            total = mt - mx.sum()
        # total = mt - sum_b max_b: the two cotangents (+1, -1 per sentence) seed the backward pass directly
        grads = torch.autograd.grad([mt, mx], leaves + ([] if with_scorer else pot), [one, minus_one.view(mx.shape)])
        return total, dict(zip(names, grads[:len(names)])), grads[len(names):]

    step.names, step.P, step.lengths, step.wiring = names, P, lengths, "r3"
    step.trainable = ("b", "b_enc", "ln_b", "ln_w", "w1", "w2", "w_enc")
    return step
