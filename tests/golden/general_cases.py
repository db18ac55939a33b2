"""The configurations of the GENERAL-engine fixtures (tests/golden/general_*.npz): UITBase geometries outside the published factories'.
Shared by make_golden.py --general (which runs them through the imported reference), the oracle tests and the GPU tests.  Data only."""

# kwargs: UITBase keyword arguments (models/uit.py:253-293) except act_layer / attention_type, which are named here ('relu' | 'gelu',
# 'bneck' | 'full') and turned into the class / string by whoever builds the model; clips: (batch, samples) per forward call
GENERAL_CASES = {
    # a wider front-end: 128 mels from a 1024-point FFT -> 8 x 6 patch tokens
    "mels128_fft1024": dict(
        kwargs=dict(outputdim=50, patch_size=16, patch_stride=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=3.0, pooling="mean",
                    target_length=102, n_mels=128, n_fft=1024, win_size=1024, hop_size=160),
        act="relu", attention="bneck", seed=1, clips=[(4, 16000), (3, 20000)]),
    # 8 x 8 patches, three heads, the UITBase defaults for attention / activation / pooling: 8 x 12 + 1 tokens
    "patch8_token": dict(
        kwargs=dict(outputdim=30, patch_size=8, patch_stride=8, embed_dim=96, depth=2, num_heads=3, mlp_ratio=2.0, pooling="token",
                    target_length=102),
        act="gelu", attention="full", seed=2, clips=[(3, 16000), (2, 16384), (2, 9000)]),
    # a 25 ms / 12.5 ms front-end without centring (n_fft 400 is no power of two: the DFT path), 40 mels of which the 16-row conv uses 32,
    # four heads of width 4, pooling 'dm', crops of 64 frames
    "fft400_dm": dict(
        kwargs=dict(outputdim=20, patch_size=16, patch_stride=16, embed_dim=64, depth=3, num_heads=4, mlp_ratio=4.0, pooling="dm",
                    target_length=64, n_mels=40, n_fft=400, win_size=400, hop_size=200, center=False, eval_avg="max"),
        act="relu", attention="bneck", seed=3, clips=[(3, 16000), (2, 9600)]),
    # a longer context: 16 time patches per crop, heads of width 32, f_min / f_max away from the defaults
    "tl256_h4": dict(
        kwargs=dict(outputdim=40, patch_size=16, patch_stride=16, embed_dim=128, depth=1, num_heads=4, mlp_ratio=4.0, pooling="token",
                    target_length=256, n_mels=64, f_min=50, f_max=7600, hop_size=128, win_size=400),
        act="gelu", attention="full", seed=4, clips=[(2, 30000), (2, 40000)]),
    # UITBase() as the reference's constructor builds it when NOTHING but the class count is given (models/uit.py:253-275): 768 wide, 12
    # blocks, 12 heads, MLP x 4, full attention, GELU, cls token, target_length 1012 -- 85.7 M parameters, 4 x 62 + 1 tokens on a 10 s clip
    "default_uitbase": dict(
        kwargs=dict(outputdim=527, patch_size=16, patch_stride=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, pooling="token",
                    target_length=1012),
        act="gelu", attention="full", seed=5, clips=[(1, 160000)]),
}
SMALL_CASES = [t for t in GENERAL_CASES if t != "default_uitbase"]      # (what the sanitizer run and the quick loops walk)


def state_dict_args(case):
    """keyword arguments of uit_mobile_amd.synth.make_general_state_dict for a case"""
    k = case["kwargs"]
    return dict(embed_dim=k["embed_dim"], depth=k["depth"], mlp_ratio=k["mlp_ratio"], attention=case["attention"], n_mels=k.get("n_mels", 64),
                n_fft=k.get("n_fft", 512), win_size=k.get("win_size", 512), f_min=float(k.get("f_min", 0)), f_max=float(k.get("f_max", 8000)),
                patch_size=k["patch_size"], target_length=k["target_length"], outputdim=k["outputdim"], seed=case["seed"])
