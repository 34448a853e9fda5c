"""Host-side boundary tests that need no GPU: the gin surface, the architecture registry, variable
naming / shapes / counts (built on the "meta" device: variables and shapes only, no arithmetic),
and the C-ABI library's exported symbols.

Reference pins: architectures/resnet_norm_test.py:30-369 (variable lists),
architectures/resnet_biggan_test.py:139,154 (BigGAN parameter totals), SURVEY.md App. B (totals of
the other example configs), App. C (gin surface), runner_lib.py:72-111 (options dict).
"""
import ctypes
import json
import os

import pytest
import torch

import __graft_entry__ as entry
from tests import gan_util as U

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.json")))
CONFIGS = sorted(f for f in os.listdir(U.CONFIG_DIR) if f.endswith(".gin"))

PARAM_TOTALS = {   # SURVEY.md App. B; the BigGAN / ResNet-CIFAR rows are reference-test pins
    "biggan_imagenet128.gin": (70433988, 87982370),
    "resnet_cifar10.gin": (5849603, 1483137),
    "dcgan_celeba64.gin": (5364739, 4314753),
    "sndcgan_celebahq128.gin": (19926019, 5983745),
    "resnet_lsun-bedroom128.gin": (13786115, 15086529),
}


def test_example_configs_are_the_reference_files():
  assert CONFIGS == ["biggan_imagenet128.gin", "dcgan_celeba64.gin", "resnet_cifar10.gin",
                     "resnet_lsun-bedroom128.gin", "sndcgan_celebahq128.gin"]


@pytest.mark.parametrize("config", CONFIGS)
def test_config_parses_and_builds(config):
  gan, options, dataset = U.build_product(config, 4, "meta")
  g = sum(v.numel() for _, v in gan.store.trainable_variables("generator"))
  d = sum(v.numel() for _, v in gan.store.trainable_variables("discriminator"))
  assert (g, d) == PARAM_TOTALS[config]
  for key in ("architecture", "batch_size", "gan_class", "lambda", "training_steps", "z_dim",
              "disc_iters"):
    assert key in options
  # every trainable variable belongs to exactly one network (modular_gan.py:345-357)
  names = [n for n, _ in gan.store.trainable_variables()]
  assert all(n.startswith("generator/") or n.startswith("discriminator/") for n in names)


def test_biggan_pins():
  pin = PINS["param_counts"]["resnet_biggan_arch_128"]
  assert PARAM_TOTALS["biggan_imagenet128.gin"] == (pin["G"], pin["D"])
  gan, options, _ = U.build_product("biggan_imagenet128.gin", 2, "meta")
  assert options["disc_iters"] == 2 and options["batch_size"] == 2048
  v = gan.store.vars
  # hierarchical z: 120 / 6 = 20 dims per chunk; CBN conditions on z chunk (20) + embed_y (128)
  assert tuple(v["generator/fc_noise/kernel"].shape) == (20, 4 * 4 * 16 * 96)
  assert tuple(v["generator/B1/bn1/condition/gamma/kernel"].shape) == (148, 1536)
  assert tuple(v["generator/embed_y/kernel"].shape) == (1000, 128)
  assert "generator/embed_y/kernel/u_var" not in v            # embed_y is not spectrally normed
  assert tuple(v["discriminator/embedding_fc/kernel"].shape) == (1000, 1536)
  # singular_value="auto": the vector sits on the smaller side (arch_ops.py:487-498)
  assert tuple(v["generator/fc_noise/kernel/u_var"].shape) == (20, 1)
  assert tuple(v["discriminator/B6/same_conv1/kernel/u_var"].shape) == (1, 1536)
  assert gan.store.initializers["generator/B1/up_conv1/kernel"] == "orthogonal"
  assert gan.store.initializers["discriminator/embedding_fc/kernel"] == "glorot_normal"
  assert gan.store.initializers["generator/non_local_block/sigma"] == "zeros"


def test_biggan_deep_pins():
  """resnet_biggan_deep_test.py:31-60: 50,244,484 generator / 34,590,210 discriminator weights at
  128 px (z_dim 128, 1000 classes, conditional batch norm), output shapes, and the naming of the
  bottleneck blocks / parameter-free shortcuts."""
  from compare_gan_amd import gin
  from compare_gan_amd.architectures import arch_ops as ops
  from compare_gan_amd.architectures import resnet_biggan_deep as deep
  gin.clear_config()
  store = ops.VariableStore("meta")
  with ops.use_store(store):
    z = torch.empty((2, 128), dtype=torch.float32, device="meta")
    y = torch.empty((2, 1000), dtype=torch.float32, device="meta")
    gen = deep.Generator(image_shape=(128, 128, 3), batch_norm_fn=ops.conditional_batch_norm)
    fake = gen(z, y=y, is_training=True)
    assert tuple(fake.shape) == (2, 128, 128, 3)
    out = deep.Discriminator()(fake, y=y, is_training=True)
    assert [tuple(o.shape) for o in out] == [(2, 1), (2, 1), (2, 2048)]
  tv = store.trainable_variables()
  pin = PINS["param_counts"]["resnet_biggan_deep_arch_128"]
  assert sum(v.numel() for n, v in tv if n.startswith("generator/")) == pin["G"]
  assert sum(v.numel() for n, v in tv if n.startswith("discriminator/")) == pin["D"]
  v = dict(tv)
  # no hierarchical z: every conditional BN sees concat(z[128], embed_y[128])
  assert tuple(v["generator/B1/conv1/bn/condition/gamma/kernel"].shape) == (256, 2048)
  assert tuple(v["generator/B4/conv2/3x3_conv/kernel"].shape) == (3, 3, 512, 512)   # max(2048,1024)/4
  assert tuple(v["generator/fc_noise/kernel"].shape) == (256, 4 * 4 * 2048)
  # G's shortcuts have no weights; D's "down" blocks append out - in channels with one 1x1 conv
  assert not any("shortcut" in n for n in v if n.startswith("generator/"))
  assert tuple(v["discriminator/B1/shortcut/add_channels/kernel"].shape) == (1, 1, 128, 128)
  assert "discriminator/B2/shortcut/add_channels/kernel" not in v
  # self-attention where the feature map is 64x64
  assert tuple(v["generator/non_local_block/conv2d_theta/kernel"].shape) == (1, 1, 256, 32)
  assert tuple(v["discriminator/non_local_block/conv2d_theta/kernel"].shape) == (1, 1, 256, 32)
  with pytest.raises(ValueError):
    with ops.use_store(ops.VariableStore("meta")):
      deep.Generator(image_shape=(48, 48, 3))(z, y=y, is_training=True)


def test_biggan_deep_oracle_and_product_agree_structurally():
  """The oracle's restatement of resnet_biggan_deep.py and the product create the same variables
  (names and shapes, spectral-norm vectors and BN accumulators included) under
  biggan_imagenet128.gin's settings -- what the GPU parity test mirrors one into the other by."""
  from oracle import architectures as OA
  from oracle import arch_ops as oops
  bind = ['options.architecture = "resnet_biggan_deep_arch"',
          "resnet_biggan_deep.Generator.ch = 32", "resnet_biggan_deep.Discriminator.ch = 32"]
  gan, options, _ = U.build_product("biggan_imagenet128.gin", 2, "meta", bindings=bind)
  assert options["architecture"] == "resnet_biggan_deep_arch"
  product = {n: tuple(v.shape) for n, v in gan.store.vars.items()}
  vs = oops.VarStore(dtype=torch.float64)
  sn = oops.SNConfig(singular_value="auto")
  g_cfg = OA.ArchConfig(batch_norm_fn="conditional_batch_norm", spectral_norm=True,
                        bn_cfg=oops.BNConfig(0.9, 1e-5, use_moving_averages=False), sn_cfg=sn,
                        embed_y=True, ch=32)
  d_cfg = OA.ArchConfig(spectral_norm=True, sn_cfg=sn, project_y=True, ch=32)
  z = torch.zeros(2, 120, dtype=torch.float64)
  y = torch.zeros(2, 1000, dtype=torch.float64)
  y[0, 3] = y[1, 977] = 1.0
  with torch.no_grad():
    img = OA.biggan_deep_generator(vs, g_cfg, z, y, True, (128, 128, 3))
    prob, logit, h = OA.biggan_deep_discriminator(vs, d_cfg, img, y, True)
  assert tuple(img.shape) == (2, 128, 128, 3) and tuple(h.shape) == (2, 512)
  assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0
  oracle = {n: tuple(v.shape) for n, v in vs.vars.items()}
  assert oracle == product


def _cifar_vars(module_kind, trainable_only=True, **kwargs):
  from compare_gan_amd import gin
  from compare_gan_amd.architectures import arch_ops as ops
  from compare_gan_amd.architectures import resnet_cifar
  gin.clear_config()
  store = ops.VariableStore("meta")
  with ops.use_store(store):
    y = torch.empty((8, 10), dtype=torch.bfloat16, device="meta")
    if module_kind == "G":
      z = torch.empty((8, 128), dtype=torch.float32, device="meta")
      out = resnet_cifar.Generator(image_shape=(32, 32, 3), **kwargs)(z, y=y, is_training=True)
      assert tuple(out.shape) == (8, 32, 32, 3)
    else:
      x = torch.empty((8, 32, 32, 3), dtype=torch.bfloat16, device="meta")
      resnet_cifar.Discriminator(**kwargs)(x, y=y, is_training=True)
  if trainable_only:
    return [[n, list(v.shape)] for n, v in store.trainable_variables()]
  return [[n, list(v.shape)] for n, v in store.global_variables()]


def test_resnet_cifar_variable_lists():
  """Names, shapes AND creation order of the reference's expected_variables lists."""
  from compare_gan_amd.architectures import arch_ops as ops
  pins = PINS["resnet_cifar_variables"]
  assert _cifar_vars("G") == pins["testDefaultGenerator"]
  assert _cifar_vars("D") == pins["testDefaultDiscriminator"]
  assert _cifar_vars("G", batch_norm_fn=ops.batch_norm) == pins["testDefaultGeneratorWithBatchNorm"]
  assert _cifar_vars("G", batch_norm_fn=ops.conditional_batch_norm) == \
      pins["testDefaultGeneratorWithConditionalBatchNorm"]
  assert _cifar_vars("G", batch_norm_fn=ops.self_modulated_batch_norm) == \
      pins["testDefaultGeneratorWithSelfModulatedBatchNorm"]
  assert _cifar_vars("G", trainable_only=False, spectral_norm=True) == \
      pins["testDefaultGeneratorWithSpectralNorm"]


def test_unknown_architecture_raises():
  gan, options, dataset = U.build_product("resnet_cifar10.gin", 2, "meta")
  from compare_gan_amd.gans import modular_gan
  params = dict(options)
  params["architecture"] = "no_such_arch"
  g = modular_gan.ModularGAN(dataset=dataset, parameters=params, model_dir="/tmp/x")
  with pytest.raises(NotImplementedError):
    g.generator  # pylint: disable=pointless-statement
  with pytest.raises(NotImplementedError):
    g.discriminator  # pylint: disable=pointless-statement


def test_conditional_without_labels_raises():
  with pytest.raises(ValueError):
    U.build_product("dcgan_celeba64.gin", 2, "meta", bindings=["ModularGAN.conditional = True"])


def test_error_conventions():
  from compare_gan_amd.architectures import arch_ops as ops
  store = ops.VariableStore("meta")
  with ops.use_store(store):
    with pytest.raises(ValueError):   # arch_ops.py:276-279
      ops.standardize_batch(torch.empty((2, 3, 4), device="meta"), is_training=True)
    with pytest.raises(ValueError):   # arch_ops.py:255-257
      ops.standardize_batch(torch.empty((2, 3), device="meta"), is_training=True, data_format="NWC")
    with pytest.raises(ValueError):   # arch_ops.py:427-430
      ops.conditional_batch_norm(torch.empty((2, 4, 4, 3), device="meta"), None, True, False)
    with pytest.raises(ValueError):   # arch_ops.py:400-401
      ops.self_modulated_batch_norm(torch.empty((2, 4, 4, 3), device="meta"), None, True, False)
    with pytest.raises(ValueError):   # arch_ops.py:470-472
      ops.spectral_norm(torch.empty((5,), device="meta"))
    with pytest.raises(ValueError):   # arch_ops.py:59-63
      ops.weight_initializer(initializer="nope")


def test_gin_surface():
  from compare_gan_amd import gin
  U.build_product("biggan_imagenet128.gin", 2, "meta")
  assert gin.query_parameter("spectral_norm.singular_value") == "auto"
  assert gin.query_parameter("standardize_batch.use_moving_averages") is False
  assert gin.query_parameter("options.lamba") == 1
  assert gin.query_parameter("resnet_biggan.Generator.hierarchical_z") is True
  s = gin.operative_config_str()
  assert "ModularGAN.g_use_ema = True" in s
  # required arguments left unbound must be reported, not silently defaulted (runner_lib.py:73-76)
  gin.clear_config()
  from compare_gan_amd import runner_lib
  with pytest.raises(Exception):
    runner_lib.get_options_dict()


def test_call_with_accepted_args():
  from compare_gan_amd import utils

  def f(a, b=2):
    return a + b
  assert utils.call_with_accepted_args(f, a=1, c=5) == 3
  assert utils.call_with_accepted_args(f, a=1, b=5, zzz=0) == 6


# -- C-ABI ---------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
  from compare_gan_amd.hip import _lib
  assert os.path.exists(_lib.LIB_PATH), "run python -m compare_gan_amd.csrc.build"
  lib = ctypes.CDLL(_lib.LIB_PATH)
  declared = entry.declared_symbols()
  assert len(declared) >= 60
  for name in declared:
    assert hasattr(lib, name), "libcgamd.so does not export %s" % name
  # the ctypes binding covers exactly the declared surface
  assert sorted(_lib.SIGNATURES) == declared


def test_library_argument_errors_without_gpu():
  """Bad-argument paths return error codes before any launch (include/cgamd.h: never throws)."""
  from compare_gan_amd.hip import _lib
  lib = _lib.load()
  assert lib.cg_abi_version() >= 1
  geom = _lib.ConvGeom(0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)
  rc = lib.cg_gconv(ctypes.byref(geom), None, None, None, 0, None, None, 0.0, None, 0.0, None, None)
  assert rc == -1 and b"cg_gconv" in lib.cg_last_error()
  geom = _lib.ConvGeom(1, 4, 4, 8, 4, 4, 8, 3, 3, 1, 3, 1, 1)    # U = 3 is not a power of two
  rc = lib.cg_gconv(ctypes.byref(geom), None, None, None, 0, None, None, 0.0, None, 0.0, None, None)
  assert rc == -2


def test_deferred_reduction_context_without_gpu():
  """cgDeferCtx is caller-owned host state (no launch until a flush has something to run): create /
  pending / abort / flush-of-nothing / destroy work without a device, contexts are independent
  objects, and the NULL context is rejected where one is required."""
  from compare_gan_amd.hip import _lib
  lib = _lib.load()
  a, b = lib.cg_defer_create(), lib.cg_defer_create()
  assert a and b and a != b
  assert lib.cg_defer_pending(a) == 0 and lib.cg_defer_pending(b) == 0
  assert lib.cg_defer_abort(a) == 0
  assert lib.cg_defer_flush(a, None) == 0           # nothing recorded: no launch
  assert lib.cg_defer_flush(None, None) == -1 and b"cg_defer_flush" in lib.cg_last_error()
  assert lib.cg_defer_abort(None) == -1
  assert lib.cg_defer_pending(None) == 0
  lib.cg_defer_destroy(a)
  lib.cg_defer_destroy(b)
  lib.cg_defer_destroy(None)


def test_weight_gradient_consumers_are_classified():
  """Fn._weight_grad_read_late: a kernel tensor whose gradient goes to the optimiser (a variable, a
  view of one, the output of a spectral-norm Function) may have its weight gradient completed at the
  flush; anything else -- the zero padding of the self-attention projections -- reads it inside the
  backward pass.  The classification rests on autograd node names of this torch version."""
  from compare_gan_amd.hip import functional as Fn
  w = torch.randn(1, 1, 8, 4, requires_grad=True)
  assert Fn._weight_grad_read_late(w)
  assert Fn._weight_grad_read_late(w.detach())                    # no gradient at all
  assert Fn._weight_grad_read_late(w.reshape(8, 4).reshape(1, 1, 8, 4))
  assert Fn._weight_grad_read_late(w.view(8, 4))
  assert not Fn._weight_grad_read_late(torch.nn.functional.pad(w, (0, 4)))
  assert not Fn._weight_grad_read_late(w * 2.0)
  assert not Fn._weight_grad_read_late(torch.nn.functional.pad(w, (0, 4)).reshape(8, 8))

  class SpectralNormStandIn(torch.autograd.Function):   # same name prefix as the product's Functions
    @staticmethod
    def forward(ctx, x):
      return x * 1.0

    @staticmethod
    def backward(ctx, g):
      return g
  assert Fn._weight_grad_read_late(SpectralNormStandIn.apply(w))
  assert Fn._weight_grad_read_late(SpectralNormStandIn.apply(w).reshape(8, 4))
  assert type(Fn.SpectralNormFn.apply).__name__   # the product's names start with "SpectralNorm"
  assert Fn.SpectralNormFn.__name__.startswith("SpectralNorm")
  assert Fn.SpectralNormBatchFn.__name__.startswith("SpectralNorm")


def test_deferred_wgrads_scopes_on_the_host():
  """Fn.deferred_wgrads (host state machine only, no device): scopes nest and restore their flags,
  reductions are recorded only while the scope says so, an inner non-deferring scope flushes first,
  an exception drops what was recorded, CPU tensors never get a context, the scaled-weight Function
  of the attention block is classified as a reader of its weight gradient inside the backward pass,
  and ops.twice_differentiable() nests."""
  from compare_gan_amd.architectures import arch_ops as ops
  from compare_gan_amd.hip import functional as Fn
  D = Fn._DEFER   # pylint: disable=protected-access
  assert (D["on"], D["reduce"]) == (False, False)
  t = torch.zeros(2)
  with Fn.deferred_wgrads(True):
    assert D["on"] and D["reduce"] == (Fn._DEFER_REDUCE and not Fn._WGRAD["enabled"])   # pylint: disable=protected-access
    assert Fn._reduce_ctx(t) is None                      # CPU tensor: no context
    with Fn.deferred_wgrads(False):
      assert (D["on"], D["reduce"]) == (False, False)
    assert D["on"]                                        # restored
    D["wptrs"].add(123)
  assert (D["on"], D["reduce"]) == (False, False) and not D["wptrs"] and not D["jobs"]
  with pytest.raises(RuntimeError):
    with Fn.deferred_wgrads(True):
      D["jobs"].append("never run")
      raise RuntimeError("backward failed")
  assert (D["on"], D["reduce"]) == (False, False) and not D["jobs"]
  w = torch.randn(1, 1, 8, 4, requires_grad=True)
  assert not Fn._weight_grad_read_late(w * torch.tensor(0.5))   # what ScaleWeightFn's output looks like
  assert not ops._DOUBLE_BWD[0]                                 # pylint: disable=protected-access
  with ops.twice_differentiable():
    assert ops._DOUBLE_BWD[0]
    with ops.twice_differentiable():
      assert ops._DOUBLE_BWD[0]
    assert ops._DOUBLE_BWD[0]
  assert not ops._DOUBLE_BWD[0]


def test_product_refuses_cpu_tensors():
  from compare_gan_amd.hip import kernels as K
  with pytest.raises(ValueError, match="no CPU fallback"):
    K.lrelu(torch.zeros(4, dtype=torch.bfloat16), 0.2)


def test_product_never_imports_the_oracle():
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  bad = []
  for dirpath, _, files in os.walk(os.path.join(root, "compare_gan_amd")):
    for f in files:
      if f.endswith(".py"):
        text = open(os.path.join(dirpath, f)).read()
        if "import oracle" in text or "from oracle" in text:
          bad.append(os.path.join(dirpath, f))
  assert not bad, bad


ARCH_SHAPES = [("infogan_arch", (28, 28, 1)), ("infogan_arch", (32, 32, 3)), ("infogan_arch", (64, 64, 3)),
               ("resnet_stl_arch", (48, 48, 1)), ("resnet_stl_arch", (48, 48, 3)),
               ("resnet30_arch", (128, 128, 3))]


def _arch_modules(arch):
  from compare_gan_amd.architectures import infogan, resnet30, resnet_stl
  return {"infogan_arch": infogan, "resnet_stl_arch": resnet_stl, "resnet30_arch": resnet30}[arch]


@pytest.mark.parametrize("arch,image_shape", ARCH_SHAPES)
def test_remaining_architectures_build_like_the_oracle(arch, image_shape):
  """architectures_test.py:30-58,76-159 (assertArchitectureBuilds) for infogan / resnet_stl /
  resnet30 on the shape-only device: output shapes, and the product creates exactly the variables
  (names, shapes) of the oracle's restatement of infogan.py:35-100, resnet_stl.py:33-108,
  resnet30.py:36-143 -- which also runs the arithmetic on the CPU: images and predictions in [0, 1]."""
  from compare_gan_amd import gin
  from compare_gan_amd.architectures import arch_ops as ops
  from oracle import architectures as OA
  from oracle import arch_ops as oops
  gin.clear_config()
  mod = _arch_modules(arch)
  bs, z_dim = 2, 128
  store = ops.VariableStore("meta")
  with ops.use_store(store):
    z = torch.empty((bs, z_dim), dtype=torch.float32, device="meta")
    gen = mod.Generator(image_shape=image_shape, batch_norm_fn=ops.batch_norm)
    fake = gen(z, y=None, is_training=True)
    assert tuple(fake.shape) == (bs,) + image_shape
    out, logit, _ = mod.Discriminator()(fake.to(torch.bfloat16), y=None, is_training=True)
    assert tuple(out.shape) == (bs, 1) and tuple(logit.shape) == (bs, 1)
  product = {n: tuple(v.shape) for n, v in store.vars.items()}
  if arch == "resnet30_arch":
    # 6 x 5 same-resolution blocks + 5 re-sampling blocks per network (resnet30.py:66-82)
    assert sum(1 for n in product if n.endswith("conv1/kernel")) == 2 * 35
  vs = oops.VarStore(dtype=torch.float64)
  g_cfg = OA.ArchConfig(batch_norm_fn="batch_norm", bn_cfg=oops.BNConfig(0.999, 1e-3))
  d_cfg = OA.ArchConfig()
  with torch.no_grad():
    img = OA.GENERATORS[arch](vs, g_cfg, torch.randn(bs, z_dim, dtype=torch.float64), None, True,
                              image_shape)
    prob, _, _ = OA.DISCRIMINATORS[arch](vs, d_cfg, img, None, True)
  assert tuple(img.shape) == (bs,) + image_shape
  assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0
  assert 0.0 <= float(prob.min()) and float(prob.max()) <= 1.0
  oracle = {n: tuple(v.shape) for n, v in vs.vars.items()}
  assert oracle == product


def test_ssgan_s3gan_heads_and_rotations():
  """gans/ssgan.py:79-102 and gans/s3gan.py:98-176 on the shape-only device: the auxiliary heads
  live under scopes that D's scope prefix matches (so they train with D), in the reference's
  creation order; and rotate_images (gans/utils.py:38-50) against the oracle's index form."""
  import numpy as np
  from compare_gan_amd.gans import s3gan, ssgan   # noqa: F401  (register SSGAN / S3GAN)
  from oracle import modular_gan as omg
  x = torch.arange(2 * 4 * 4 * 3, dtype=torch.float32).reshape(2, 4, 4, 3)
  assert torch.equal(ssgan.rotate_images(x, (1, 2, 3)), omg.rotate_images(x, (1, 2, 3)))
  assert torch.equal(ssgan.rotate_images(x), omg.rotate_images(x))
  assert np.array_equal(ssgan.rotate_images(x, (1,)).numpy(), np.rot90(x.numpy(), 1, axes=(1, 2)))
  gan, _, _ = U.build_product("resnet_cifar10.gin", 4, "meta", bindings=(
      "options.gan_class = @SSGAN", "SSGAN.rotated_batch_size = 4"))
  heads = [n for n, _ in gan.store.trainable_variables("discriminator")
           if not n.startswith("discriminator/")]
  assert heads == ["discriminator_rotation/score_classify/kernel",
                   "discriminator_rotation/score_classify/bias"]
  assert tuple(gan.store.vars[heads[0]].shape) == (128, 4)
  with pytest.raises(Exception):      # rotated_batch_size is gin.REQUIRED (ssgan.py:48)
    U.build_product("resnet_cifar10.gin", 4, "meta", bindings=("options.gan_class = @SSGAN",))
  gan, _, _ = U.build_product("biggan_imagenet128.gin", 8, "meta", bindings=(
      "options.gan_class = @S3GAN", "S3GAN.rotated_batch_fraction = 2", "S3GAN.use_predictor = True",
      "S3GAN.project_y = True", "resnet_biggan.Generator.ch = 16",
      "resnet_biggan.Discriminator.ch = 16"))
  heads = [n for n, _ in gan.store.trainable_variables("discriminator")
           if not n.startswith("discriminator/")]
  assert heads == ["discriminator_rotation/score_classify/kernel",
                   "discriminator_rotation/score_classify/bias",
                   "discriminator_predictor/predictor_linear/kernel",
                   "discriminator_predictor/predictor_linear/bias",
                   "discriminator_projection/kernel"]
  assert tuple(gan.store.vars["discriminator_projection/kernel"].shape) == (1000, 16 * 16)
  with pytest.raises(ValueError):     # s3gan.py:80-81
    U.build_product("biggan_imagenet128.gin", 8, "meta", bindings=(
        "options.gan_class = @S3GAN", "S3GAN.rotated_batch_fraction = 2",
        "S3GAN.use_predictor = True", "S3GAN.project_y = False"))


def test_joint_generation_policy():
  """ModularGAN.joint_generation_groups(): the generator forwards of the discriminator sub-steps
  are batched with per-sub-step statistics only when that is the arithmetic of the separate calls
  (modular_gan.py:464-467) -- not for a spectrally normalised generator (one power iteration per
  call, arch_ops.py:479-535) -- and jointly (one statistics group) under the reference's
  experimental_joint_gen_for_disc (modular_gan.py:444-458)."""
  gan, _, _ = U.build_product("resnet_cifar10.gin", 2, "meta")
  assert gan.joint_generation_groups() == 5                       # disc_iters = 5, G without SN
  gan, _, _ = U.build_product("resnet_cifar10.gin", 2, "meta", bindings=("options.disc_iters = 1",))
  assert gan.joint_generation_groups() is None
  gan, _, _ = U.build_product("resnet_cifar10.gin", 2, "meta",
                              bindings=("ModularGAN.experimental_joint_gen_for_disc = True",))
  assert gan.joint_generation_groups() == 1
  bind = ("resnet_biggan.Generator.ch = 16", "resnet_biggan.Discriminator.ch = 16")
  gan, _, _ = U.build_product("biggan_imagenet128.gin", 2, "meta", bindings=bind)
  assert gan.joint_generation_groups() is None                    # G.spectral_norm = True
  gan, _, _ = U.build_product("biggan_imagenet128.gin", 2, "meta", bindings=bind + (
      "ModularGAN.experimental_joint_gen_for_disc = True",))
  assert gan.joint_generation_groups() == 1
  with pytest.raises(ValueError):     # modular_gan.py:538-540
    gan.train_step_not_unrolled(torch.empty((2, 128, 128, 3)), torch.empty((2,), dtype=torch.int32))


def test_small_conv_swizzle_is_conflict_free():
    """The LDS swizzles of cg_conv_small.hip (sconv_kernel): for every filter tap, both pixel halves
    and every 16-byte k chunk, the 16 lanes of each ds_read_b128 service group (MI355X_MICROARCH.md,
    LDS table: {0-3,12-15,20-27}, {4-11,16-19,28-31} of a half wave) must hit 16 distinct 16-byte
    bank groups of the 256-byte LDS row.  Window rows are 128 B (row r -> bank groups 8 (r % 2) + c),
    weight-unit rows are 64 B.  hconv_kernel's multi-image tiles (cg_conv_halo.hip, TWL == 3: four
    8x8 images with 10x10 windows) use the 8x8 swizzle and pixel order checked here."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    for tw, swz in ((8, lambda il, y, x: ((x >> 1) & 3) | ((y & 1) << 2)),
                    (4, lambda il, y, x: ((x >> 1) & 1) | ((y & 1) << 1) | ((il & 1) << 2))):
        pitch, img_rows = tw + 2, (tw + 2) * (tw + 2)
        for i in range(2):
            for ri in range(3):
                for si in range(3):
                    for grp in groups:
                        for chunk in range(8):
                            seen = set()
                            for lane in grp:
                                p = i * 32 + lane
                                il, y, x = p // (tw * tw), (p // tw) % tw, p % tw
                                yy, xx = y + ri, x + si
                                row = il * img_rows + yy * pitch + xx
                                seen.add((row % 2) * 8 + ((chunk ^ swz(il, yy, xx)) & 7))
                            assert len(seen) == 16, (tw, i, ri, si, chunk)
    for grp in groups:
        for chunk in range(4):
            seen = {((row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)) // 16) % 16 for row in grp}
            assert len(seen) == 16
