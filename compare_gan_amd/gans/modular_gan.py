"""The GAN training step on one MI355X (or one rank of a data-parallel job).

Reference: compare_gan/gans/modular_gan.py:56-670 (ModularGAN: architecture table, input split,
G forward, D step(s), G step, optimizers, EMA) on top of abstract_gan.py:29-92.  The reference
builds a TF1 graph for TPUEstimator; here the same step is executed directly: every arithmetic op
is a HIP kernel (compare_gan_amd.hip), torch.autograd only records the graph, the optimiser is one
fused multi-tensor TF-Adam(+EMA) launch per network, and the data-parallel gradient mean is one
RCCL all-reduce of a flat bucket per network (the reference's CrossShardOptimizer,
modular_gan.py:606-616).

Step semantics = the reference's *unrolled* graph (modular_gan.py:533-584):
`disc_iters` discriminator sub-steps, each on a fresh sub-batch (real images, z, sampled labels)
and a fresh G forward, followed by one generator sub-step on another fresh sub-batch.
"""
import os

import torch

from compare_gan_amd import gin
from compare_gan_amd import utils
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.architectures import dcgan
from compare_gan_amd.architectures import infogan
from compare_gan_amd.architectures import resnet30
from compare_gan_amd.architectures import resnet5
from compare_gan_amd.architectures import resnet_biggan
from compare_gan_amd.architectures import resnet_biggan_deep
from compare_gan_amd.architectures import resnet_cifar
from compare_gan_amd.architectures import resnet_stl
from compare_gan_amd.architectures import sndcgan
from compare_gan_amd.gans import consts as c
from compare_gan_amd.gans import loss_lib
from compare_gan_amd.gans import penalty_lib
from compare_gan_amd.gans.abstract_gan import AbstractGAN
from compare_gan_amd.hip import functional as Fn
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.tpu import tpu_ops
from compare_gan_amd.tpu import tpu_random


# ------------------------------------------------------------------------------------------------
# external configurables the example configs bind (main.py:38-39 gin.tf.external_configurables)
# ------------------------------------------------------------------------------------------------
class AdamOptimizer(object):
  """Hyper-parameters of tf.train.AdamOptimizer; the update itself is cg_adam_multi."""

  def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name="Adam"):
    self.learning_rate = learning_rate
    self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
    self.name = name


AdamOptimizer = gin.external_configurable(AdamOptimizer, name="AdamOptimizer", module="tf.train")


def _random_normal(shape, mean=0.0, stddev=1.0, name=None, device=None):
  return tpu_random.normal(shape, name or "random_normal", mean, stddev, device)


def _random_uniform(shape, minval=0.0, maxval=1.0, name=None, device=None):
  return tpu_random.uniform(shape, name or "random_uniform", minval, maxval, device)


random_normal = gin.external_configurable(_random_normal, name="normal", module="tf.random")
random_uniform = gin.external_configurable(_random_uniform, name="uniform", module="tf.random")


# Where the bucket's all-reduce + update run: on a communication stream (overlapping the next
# generator forward) for buckets of at least _DP_OVERLAP_MIN_BYTES, on the main stream below that
# -- a forked branch costs two cross-stream dependencies per update (measured on a one-rank RCCL
# group, resnet_cifar10: 6 forks add 0.6 ms to a 10.2 ms step), more than a 4 MiB all-reduce
# takes.  CGAMD_DP_OVERLAP=1 / 0 forces it on / off.
_DP_OVERLAP = os.environ.get("CGAMD_DP_OVERLAP", "auto")
_JOINT_G = os.environ.get("CGAMD_JOINT_G", "1") != "0"   # batched generator forwards (A/B switch)
_DEFER_WGRAD = os.environ.get("CGAMD_DEFER_WGRAD", "1") != "0"   # grouped small-map weight gradients
_DP_OVERLAP_MIN_BYTES = 32 << 20
# With the overlap on, the gradients leave in _DP_BUCKETS buckets (each at least
# _DP_BUCKET_MIN_BYTES) DURING the backward pass that produces them: a bucket's all-reduce starts
# as soon as its last gradient exists (tensor hooks), the last layers first -- see plan_buckets().
_DP_BUCKETS = int(os.environ.get("CGAMD_DP_BUCKETS", "2"))
_DP_BUCKET_MIN_BYTES = int(os.environ.get("CGAMD_DP_BUCKET_MIN_MB", "16")) << 20


def plan_buckets(numels, num_buckets, min_bytes):
  """Splits the variables of a network (creation = forward order, `numels` elements each, fp32)
  into at most `num_buckets` CONTIGUOUS index ranges of about equal bytes, none smaller than
  `min_bytes` -- returned in LAUNCH order: the backward pass produces the gradients of the last
  layers first, so the first bucket to leave is the tail of the list.  Contiguous ranges keep each
  bucket one slice of the flat gradient buffer (one collective per bucket, one multi-tensor Adam
  over the whole buffer afterwards).  xGMI rings are per-link bound, so the buckets stay large:
  two ~equal halves hide half of the all-reduce behind the rest of the backward pass at the cost
  of one extra ring latency."""
  total = 4 * sum(numels)
  n = max(1, min(int(num_buckets), int(total // max(1, min_bytes)), len(numels)))
  target = total / n
  ranges, hi, acc = [], len(numels), 0
  for i in range(len(numels) - 1, -1, -1):
    acc += 4 * numels[i]
    if len(ranges) < n - 1 and acc >= target and i > 0:
      ranges.append((i, hi))
      hi, acc = i, 0
  ranges.append((0, hi))
  return ranges
_NCCL_WATCHDOG_POLL_S = 0.1   # ProcessGroupNCCL's watchdog poll period (see _capture)


class _OptimizerState(object):
  """Adam slots (+ EMA shadows) for one network and the fused update."""

  def __init__(self, named_vars, opt, with_ema, device):
    self.names = [n for n, _ in named_vars]
    self.params = [v for _, v in named_vars]
    self.opt = opt
    self.m = [torch.zeros_like(v, requires_grad=False) for v in self.params]
    self.v = [torch.zeros_like(v, requires_grad=False) for v in self.params]
    self.ema = [v.detach().clone() for v in self.params] if with_ema else None
    self.table = None
    self.zero_grads = {}
    self.captured_tables = []
    self.reserved = []
    self.flat = None
    self.device = device
    self._comm = None
    self._inflight = None
    self._buckets = None       # plan_buckets() ranges, launch order
    self._armed = None         # per-backward state of the bucketed all-reduce (see arm())
    self._hooks = None

  # -- gradient buckets leaving during the backward pass ----------------------------------------------
  def _make_flat(self):
    if self.flat is None:
      self.flat = torch.empty(sum(p.numel() for p in self.params), dtype=torch.float32,
                              device=self.device)
      self.flat_views, self.flat_offsets = [], []
      off = 0
      for p in self.params:
        self.flat_views.append(self.flat[off:off + p.numel()].view(p.shape))
        self.flat_offsets.append(off)
        off += p.numel()

  def _overlap(self):
    self._make_flat()
    return _DP_OVERLAP == "1" or (_DP_OVERLAP == "auto" and
                                  self.flat.numel() * 4 >= _DP_OVERLAP_MIN_BYTES)

  def arm(self):
    """Call right before the backward pass whose gradients apply_gradients() will consume.  Data
    parallel with the overlap on: from now on each gradient reports in through a tensor hook, and a
    bucket whose gradients are all there is flattened and all-reduced on the communication stream
    while the backward pass continues on the main stream.  Otherwise a no-op.

    Collective ordering across ranks (RCCL requires every rank to issue the collectives of one
    communicator in the same order).  Two threads issue collectives: the main thread (cross-replica
    batch norm in the forward passes, tpu_ops.SyncMoments) and autograd's device worker thread (the
    batch norms' backward all-reduces and these bucket all-reduces).  They never issue concurrently:
    the main thread sits inside torch.autograd.backward()/grad() for the whole backward pass, and the
    worker issues nothing outside one.  Within a backward pass the engine runs ready nodes by
    descending sequence number -- a pure function of the order in which the forward pass created them,
    i.e. of the model code, identical on every rank -- so hooks fire, buckets fill (plan_buckets is a
    function of the parameter sizes alone) and batch-norm backwards interleave in the same order
    everywhere.  What would break it: rank-dependent control flow in a model (none in the
    architectures here), or autograd's multi-device threading (one process drives one GPU)."""
    self._armed = None
    if not (tpu_ops.data_parallel() and tpu_ops.thread_state() is None and _DP_BUCKETS > 1):
      return
    if not self._overlap():
      return
    if self._buckets is None:
      self._buckets = plan_buckets([p.numel() for p in self.params], _DP_BUCKETS,
                                   _DP_BUCKET_MIN_BYTES)
    if len(self._buckets) < 2:
      return
    if self._hooks is None:
      self._hooks = [p.register_hook(lambda g, i=i: self._on_grad(i, g))
                     for i, p in enumerate(self.params)]
    self.join()                  # the buffer of the previous update must have been consumed
    owner = [0] * len(self.params)
    for b, (lo, hi) in enumerate(self._buckets):
      for i in range(lo, hi):
        owner[i] = b
    self._armed = {"grads": [None] * len(self.params), "owner": owner,
                   "missing": [hi - lo for lo, hi in self._buckets],
                   "launched": [False] * len(self._buckets), "order": []}

  def _on_grad(self, i, grad):
    st = self._armed
    if st is None or st["grads"][i] is not None:
      return None
    st["grads"][i] = grad
    b = st["owner"][i]
    st["missing"][b] -= 1
    if st["missing"][b] == 0 and not st["launched"][b]:
      self._launch_bucket(b, st["grads"])
    return None

  def _launch_bucket(self, b, grads):
    """flatten + all-reduce of bucket b on the communication stream, behind everything enqueued so
    far on the stream that produced its gradients (and the weight-gradient side stream)."""
    st = self._armed
    lo, hi = self._buckets[b]
    Fn.join_wgrad_stream()      # deferred / side-stream weight gradients of this bucket
    main, comm = torch.cuda.current_stream(), self._comm_stream()
    comm.wait_stream(main)
    a, z = self.flat_offsets[lo], self.flat_offsets[hi - 1] + self.params[hi - 1].numel()
    with torch.cuda.stream(comm):
      K.flatten_multi([g.contiguous() for g in grads[lo:hi]], self.flat[a:z])
      tpu_ops.cross_replica_sum_(self.flat[a:z])
    st["launched"][b] = True
    st["order"].append(b)
    self._inflight = True

  def disarm(self):
    """Drops an armed state that apply_gradients() will not consume (the backward pass raised): the
    gradient references go, and so does the in-flight mark of buckets that already left (the
    ordering argument for the collectives is in arm())."""
    if self._armed is not None and self._armed["order"]:
      self.join()
    self._armed = None

  def _ensure(self, grads):
    if torch.cuda.is_current_stream_capturing():
      # every captured update keeps its own table: its gradient tensors live at addresses that
      # are fixed for all replays, and the table is only read when the graph runs.  The table
      # bytes come from reserve_tables() (allocated outside the capture, never recycled).
      if not self.reserved:
        raise RuntimeError("call reserve_tables() before capturing an optimiser update")
      t = K.AdamTable([p.detach() for p in self.params], grads, self.m, self.v, self.ema,
                      table=self.reserved.pop())
      self.captured_tables.append(t)
      return t
    if self.table is None:
      self.table = K.AdamTable([p.detach() for p in self.params], grads, self.m, self.v, self.ema)
    else:
      self.table.set_grads(grads)
    return self.table

  def reserve_tables(self, n):
    """Device buffers for the Adam tables of `n` captured updates (see AdamTable)."""
    nbytes = K.AdamTable.table_bytes(len(self.params))
    self.reserved = [torch.empty(nbytes, dtype=torch.uint8, device=self.device) for _ in range(n)]

  def apply_gradients(self, step, ema_decay=0.0, ema_start=0, grads=None):
    """All-reduce (data parallel) + fused Adam(+EMA) + step += 1.  step: device int64 counter.
    grads: one tensor per variable (defaults to the variables' .grad fields).

    Data parallel: the network's gradients are gathered into ONE flat fp32 bucket (xGMI rings are
    per-link bound: one large message per network beats many small ones), summed across replicas
    and applied with scale 1/world -- all three on the communication stream, so that work which
    does not read this network's weights (the next sub-step's generator forward after a D update)
    overlaps with the collective; join() is the dependency edge for the next reader."""
    Fn.join_wgrad_stream()   # weight gradients may still be in flight on the side stream
    if grads is None:
      grads = [p.grad for p in self.params]
    grads = list(grads)
    for i, (n, g) in enumerate(zip(self.names, grads)):
      if g is None:
        raise RuntimeError("variable %s received no gradient" % n)
      grads[i] = g.contiguous()
    o = self.opt
    ema = ema_decay if self.ema is not None else 0.0
    if not tpu_ops.data_parallel():
      self._ensure(grads).adam(o.learning_rate, o.beta1, o.beta2, o.epsilon, 1.0, step,
                               ema_decay=ema, ema_start=ema_start)
      K.counter_add(step, 1)
      return
    world = tpu_ops.num_replicas()
    self._make_flat()
    st, self._armed = self._armed, None
    if st is not None:
      # buckets left during the backward pass (arm()); the ones that could not -- a variable
      # without gradient never reports in -- leave now, then the update follows on the same stream
      self._armed = st
      for b in range(len(self._buckets)):
        if not st["launched"][b]:
          self._launch_bucket(b, grads)
      self._armed = None
      self.last_bucket_order = st["order"]
      # the update writes the variables: behind the last main-stream kernel that reads them
      self._comm_stream().wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(self._comm_stream()):
        self._ensure(self.flat_views).adam(o.learning_rate, o.beta1, o.beta2, o.epsilon,
                                           1.0 / world, step, ema_decay=ema, ema_start=ema_start)
        K.counter_add(step, 1)
      self._inflight = grads
      return
    self.join()                  # the bucket of the previous update must have been consumed
    main = torch.cuda.current_stream()
    comm = self._comm_stream() if self._overlap() else main
    if comm is not main:
      comm.wait_stream(main)
    with torch.cuda.stream(comm):
      K.flatten_multi(grads, self.flat)          # one bucket per network
      tpu_ops.cross_replica_sum_(self.flat)      # CrossShardOptimizer: gradient mean
      self._ensure(self.flat_views).adam(o.learning_rate, o.beta1, o.beta2, o.epsilon,
                                         1.0 / world, step, ema_decay=ema, ema_start=ema_start)
      K.counter_add(step, 1)     # same stream as its reader (the update above)
    if comm is not main:
      # the gradient tensors were allocated on the main stream: keep them alive until the main
      # stream has waited for the communication stream (join), or the allocator may hand their
      # memory to a later main-stream kernel while the flatten still reads it
      self._inflight = grads

  def _comm_stream(self):
    if self._comm is None:
      self._comm = torch.cuda.Stream(device=self.device)
    return self._comm

  def join(self):
    """Makes the current stream wait for an update in flight on the communication stream: called
    before anything reads this network's variables (its forward pass, EMA readers, checkpoints)
    and at the end of a step (a hipGraph capture has to rejoin every forked stream)."""
    if self._inflight is not None:
      torch.cuda.current_stream().wait_stream(self._comm)
      self._inflight = None


@gin.configurable(blacklist=["dataset", "parameters", "model_dir"])
class ModularGAN(AbstractGAN):
  """Gin-configurable GAN trainer with the reference's constructor surface."""

  def __init__(self, dataset, parameters, model_dir, deprecated_split_disc_calls=False,
               experimental_joint_gen_for_disc=False, experimental_force_graph_unroll=False,
               g_use_ema=False, ema_decay=0.9999, ema_start_step=40000,
               g_optimizer_fn=AdamOptimizer, d_optimizer_fn=None, g_lr=0.0002, d_lr=None,
               conditional=False, fit_label_distribution=False):
    super(ModularGAN, self).__init__(dataset=dataset, parameters=parameters, model_dir=model_dir)
    self._deprecated_split_disc_calls = deprecated_split_disc_calls
    self._experimental_joint_gen_for_disc = experimental_joint_gen_for_disc
    self._experimental_force_graph_unroll = experimental_force_graph_unroll
    self._g_use_ema = g_use_ema
    self._ema_decay = ema_decay
    self._ema_start_step = ema_start_step
    self._g_optimizer_fn = g_optimizer_fn
    self._d_optimizer_fn = d_optimizer_fn if d_optimizer_fn is not None else g_optimizer_fn
    self._g_lr = g_lr
    self._d_lr = g_lr if d_lr is None else d_lr
    if conditional and not self._dataset.num_classes:
      raise ValueError("Option 'conditional' selected but dataset {} does not have "
                       "labels".format(self._dataset.name))
    self._conditional = conditional
    self._fit_label_distribution = fit_label_distribution
    # Parameters that have not been ported to Gin (modular_gan.py:142-149).
    self._architecture = parameters["architecture"]
    self._z_dim = parameters["z_dim"]
    self._lambda = parameters["lambda"]
    self._disc_iters = parameters.get("disc_iters", 1)
    self.d_loss = None
    self.g_loss = None
    self.penalty_loss = None
    self._discriminator = None
    self._generator = None
    self.store = None
    self._built = False

  # -- architecture table (modular_gan.py:169-213) ---------------------------------------------------
  @property
  def conditional(self):
    return self._conditional

  @property
  def generator(self):
    if self._generator is None:
      architecture_fns = {
          c.DCGAN_ARCH: dcgan.Generator,
          c.INFOGAN_ARCH: infogan.Generator,
          c.RESNET5_ARCH: resnet5.Generator,
          c.RESNET30_ARCH: resnet30.Generator,
          c.RESNET_STL_ARCH: resnet_stl.Generator,
          c.RESNET_BIGGAN_ARCH: resnet_biggan.Generator,
          c.RESNET_BIGGAN_DEEP_ARCH: resnet_biggan_deep.Generator,
          c.RESNET_CIFAR_ARCH: resnet_cifar.Generator,
          c.SNDCGAN_ARCH: sndcgan.Generator,
      }
      if self._architecture not in architecture_fns:
        raise NotImplementedError(
            "Generator architecture {} not implemented.".format(self._architecture))
      self._generator = architecture_fns[self._architecture](
          image_shape=self._dataset.image_shape)
    return self._generator

  @property
  def discriminator(self):
    if self._discriminator is None:
      architecture_fns = {
          c.DCGAN_ARCH: dcgan.Discriminator,
          c.INFOGAN_ARCH: infogan.Discriminator,
          c.RESNET5_ARCH: resnet5.Discriminator,
          c.RESNET30_ARCH: resnet30.Discriminator,
          c.RESNET_STL_ARCH: resnet_stl.Discriminator,
          c.RESNET_BIGGAN_ARCH: resnet_biggan.Discriminator,
          c.RESNET_BIGGAN_DEEP_ARCH: resnet_biggan_deep.Discriminator,
          c.RESNET_CIFAR_ARCH: resnet_cifar.Discriminator,
          c.SNDCGAN_ARCH: sndcgan.Discriminator,
      }
      if self._architecture not in architecture_fns:
        raise NotImplementedError(
            "Discriminator architecture {} not implemented.".format(self._architecture))
      self._discriminator = architecture_fns[self._architecture]()
    return self._discriminator

  # -- graph construction ----------------------------------------------------------------------------
  def build(self, batch_size, device, seed=0):
    """Creates every variable (one shape-only pass over G and D, no kernel runs), the optimiser
    slots, EMA shadows and step counters.  batch_size is the per-replica sub-step batch."""
    self.device = torch.device(device)
    self.batch_size = batch_size
    self.seed = seed
    self.store = ops.VariableStore(self.device, seed=seed)
    meta = "meta"
    with ops.use_store(self.store):
      z = torch.empty((batch_size, self._z_dim), dtype=torch.float32, device=meta)
      y = None
      if self.conditional:
        y = torch.empty((batch_size, self._dataset.num_classes), dtype=torch.bfloat16,
                        device=meta)
      images = self.generator(z, y=y, is_training=True)
      x = torch.empty((2 * batch_size,) + tuple(images.shape[1:]), dtype=torch.bfloat16,
                      device=meta)
      y2 = None if y is None else torch.empty((2 * batch_size, y.shape[1]), dtype=y.dtype,
                                              device=meta)
      d_out = self.discriminator(x, y=y2, is_training=True)
      self._build_heads(x, d_out)     # subclasses create the variables of their extra heads
      self._check_variables()
    if self.device.type == "meta":
      self._built = True
      return self
    self.global_step = torch.zeros((), dtype=torch.int64, device=self.device)
    self.global_step_disc = torch.zeros((), dtype=torch.int64, device=self.device)
    g_vars = self.store.trainable_variables(self.generator.name)
    d_vars = self.store.trainable_variables(self.discriminator.name)
    self.g_opt = _OptimizerState(g_vars, self.get_gen_optimizer(), self._g_use_ema, self.device)
    self.d_opt = _OptimizerState(d_vars, self.get_disc_optimizer(), False, self.device)
    tpu_random.set_random_offset(seed, self.global_step)
    # a forward pass of either network first waits for an update of ITS variables that may still
    # be in flight on the communication stream (ops.prepare_module runs these before anything reads
    # a variable): subclasses and new call sites need no join() of their own
    self.store.before_call = {self.generator.name: self.g_opt.join,
                              self.discriminator.name: self.d_opt.join}
    self._built = True
    return self

  def _build_heads(self, x, d_out):
    """Shape-only hook of build(): x is the (meta) discriminator input [2B,H,W,C], d_out the
    discriminator's (prob, logits, features)."""
    del x, d_out

  def _check_variables(self):
    """Every trainable variable belongs to exactly one of G / D (modular_gan.py:345-357)."""
    t_vars = set(n for n, _ in self.store.trainable_variables())
    g_vars = set(n for n, _ in self.store.trainable_variables(self.generator.name))
    d_vars = set(n for n, _ in self.store.trainable_variables(self.discriminator.name))
    shared = g_vars & d_vars
    if shared:
      raise ValueError("Shared trainable variables: %s" % shared)
    unused = t_vars - g_vars - d_vars
    if unused:
      raise ValueError("Unused trainable variables: %s" % unused)

  def get_disc_optimizer(self, use_tpu=True):
    del use_tpu   # the cross-replica mean lives in _OptimizerState.apply_gradients
    return self._d_optimizer_fn(self._d_lr, name="d_opt")

  def get_gen_optimizer(self, use_tpu=True):
    del use_tpu
    return self._g_optimizer_fn(self._g_lr, name="g_opt")

  # -- inputs ------------------------------------------------------------------------------------------
  def _get_one_hot_labels(self, labels):
    if not self.conditional:
      raise ValueError("_get_one_hot_labels() called but GAN is not conditional.")
    return K.one_hot(labels, self._dataset.num_classes)

  @gin.configurable("z", blacklist=["shape", "name"])
  def z_generator(self, shape, distribution_fn=random_uniform, minval=-1.0, maxval=1.0,
                  stddev=1.0, name=None):
    """Random noise for G (modular_gan.py:365-384)."""
    return utils.call_with_accepted_args(distribution_fn, shape=shape, minval=minval,
                                         maxval=maxval, stddev=stddev, name=name,
                                         device=self.device)

  def label_generator(self, shape, name=None):
    """Uniform labels (modular_gan.py:386-391)."""
    if not self.conditional:
      raise ValueError("label_generator() called but GAN is not conditional.")
    return tpu_random.labels(int(shape[0]), self._dataset.num_classes, name or "sampled_labels",
                             self.device)

  def _preprocess(self, images, labels, sub_step):
    """Feature dictionary of one sub-step (modular_gan.py:393-408): real images, z, sampled
    labels; names are unique per sub-step so every sub-step draws fresh noise."""
    features = {"images": images, "_sub_step": sub_step,
                "z": self.z_generator([images.shape[0], self._z_dim], name="z/%d" % sub_step)}
    if self.conditional:
      if self._fit_label_distribution:
        features["sampled_labels"] = labels
      else:
        features["sampled_labels"] = self.label_generator(
            [images.shape[0]], name="sampled_labels/%d" % sub_step)
    return features, labels

  # -- losses (modular_gan.py:618-670) ---------------------------------------------------------------
  def create_loss(self, features, labels, params=None, is_training=True):
    """Sets self.d_loss / self.g_loss (/ self.penalty_loss) for one sub-step."""
    del params
    images = features["images"]        # real, fp32 [B,H,W,C] in [0,1]
    generated = features["generated"]  # fake, fp32
    if self.conditional:
      y = self._get_one_hot_labels(labels)
      sampled_y = self._get_one_hot_labels(features["sampled_labels"])
      all_y = torch.cat([y, sampled_y], dim=0)
    else:
      y = sampled_y = all_y = None
    a, b = getattr(self.discriminator, "input_affine", (1.0, 0.0))
    self.d_opt.join()    # a D update may still be in flight on the communication stream
    if self._deprecated_split_disc_calls:
      d_real, d_real_logits, _ = self.discriminator(
          Fn.stage_images(images, None, a, b), y=y, is_training=is_training)
      d_fake, d_fake_logits, _ = self.discriminator(
          Fn.stage_images(None, generated, a, b), y=sampled_y, is_training=is_training)
    else:
      all_images = Fn.stage_images(images, generated, a, b)   # concat + bf16 (+ input affine)
      d_all, d_all_logits, _ = self.discriminator(all_images, y=all_y, is_training=is_training)
      bsz = images.shape[0]
      d_real, d_fake = d_all[:bsz], d_all[bsz:]
      d_real_logits, d_fake_logits = d_all_logits[:bsz], d_all_logits[bsz:]
      if not d_all_logits.is_meta:
        d_real_logits._cg_joined = d_fake_logits._cg_joined = d_all_logits   # loss_lib._run
    self.d_loss, _, _, self.g_loss = loss_lib.get_losses(
        d_real=d_real, d_fake=d_fake, d_real_logits=d_real_logits, d_fake_logits=d_fake_logits)
    self.penalty_loss = None
    tpu_random.set_sub_step(features.get("_sub_step", 0))
    if torch.is_grad_enabled() and not features.get("_generator_step", False):
      penalty_loss = penalty_lib.get_penalty_loss(
          x=images, x_fake=generated, y=y, is_training=is_training,
          discriminator=self.discriminator)
      if penalty_loss is not None:
        self.penalty_loss = penalty_loss
        self.d_loss = Fn.add_f32(self.d_loss.reshape(1), penalty_loss.reshape(1), 1.0,
                                 float(self._lambda)).reshape(())

  # -- training step -------------------------------------------------------------------------------
  def _zero_grads(self, opt_state):
    for p in opt_state.params:
      p.grad = None

  def _train_discriminator(self, features, labels):
    """One D update (modular_gan.py:471-485)."""
    features = dict(features)
    features["generated"] = features["generated"].detach()
    self._set_requires_grad(self.g_opt, False)
    self._set_requires_grad(self.d_opt, True)
    with ops.use_store(self.store):
      self.create_loss(features, labels)
    # torch.autograd.grad hands the gradients over directly: no AccumulateGrad nodes, whose
    # stream affinity would break hipGraph capture (they run on the stream they were created on)
    self.d_opt.arm()
    try:
      with Fn.deferred_wgrads(self._wgrads_deferrable() and self.penalty_loss is None):
        grads = torch.autograd.grad(self.d_loss, self.d_opt.params, grad_outputs=self._unit_grad(),
                                    allow_unused=True)
    except BaseException:
      self.d_opt.disarm()   # a failed backward must not leave hooks armed / gradients referenced
      raise
    self.d_opt.apply_gradients(self.global_step_disc, grads=self._fill_unused(self.d_opt, grads))
    self.d_loss = self.d_loss.detach()
    if self.penalty_loss is not None:
      self.penalty_loss = self.penalty_loss.detach()
    self.g_loss = self.g_loss.detach()
    return self.d_loss

  def _train_generator(self, features, labels, shared_forward=False):
    """One G update (+ EMA) (modular_gan.py:487-510); D's weights get no gradient.
    shared_forward: features["generated"] already holds this sub-step's generator forward WITH
    its autograd graph (the not-unrolled step builds one forward for both updates)."""
    features = dict(features)
    features["_generator_step"] = True
    self._set_requires_grad(self.d_opt, False)
    self._set_requires_grad(self.g_opt, True)
    with ops.use_store(self.store):
      if not shared_forward:
        sampled_y = None
        if self.conditional:
          sampled_y = self._get_one_hot_labels(features["sampled_labels"])
        self.g_opt.join()
        features["generated"] = self.generator(features["z"], y=sampled_y, is_training=True)
      self.create_loss(features, labels)
    self.g_opt.arm()
    try:
      with Fn.deferred_wgrads(self._wgrads_deferrable()):
        grads = torch.autograd.grad(self.g_loss, self.g_opt.params, grad_outputs=self._unit_grad(),
                                    allow_unused=True)
    except BaseException:
      self.g_opt.disarm()
      raise
    self.g_opt.apply_gradients(self.global_step, ema_decay=self._ema_decay,
                               ema_start=self._ema_start_step,
                               grads=self._fill_unused(self.g_opt, grads))
    self._set_requires_grad(self.d_opt, True)
    self.g_loss = self.g_loss.detach()
    self.d_loss = self.d_loss.detach()
    return self.g_loss

  def _wgrads_deferrable(self):
    """Weight gradients may be deferred and grouped (Fn.deferred_wgrads) when every weight gets
    exactly ONE gradient contribution in the backward pass: one discriminator / generator call per
    graph (the base class's create_loss without deprecated_split_disc_calls) and no penalty term
    (checked by the caller: a penalty differentiates the discriminator a second time)."""
    return (_DEFER_WGRAD and not self._deprecated_split_disc_calls and
            type(self).create_loss is ModularGAN.create_loss)

  def _unit_grad(self):
    """d(loss)/d(loss) = 1, allocated once (autograd would launch a fill per backward pass)."""
    if getattr(self, "_unit", None) is None:
      self._unit = torch.ones((), dtype=torch.float32, device=self.device)
    return self._unit

  @staticmethod
  def _fill_unused(opt_state, grads):
    """Variables the loss does not depend on get a zero gradient (tf.gradients returns None and
    Adam skips them; with zero m/v a zero gradient is the same no-op, e.g. the attention block's
    theta/phi kernels while sigma == 0 still receive exact zeros through the graph)."""
    out = []
    for n, p, g in zip(opt_state.names, opt_state.params, grads):
      if g is None:
        if n not in opt_state.zero_grads:
          opt_state.zero_grads[n] = torch.zeros_like(p, requires_grad=False)
        g = opt_state.zero_grads[n]
      out.append(g)
    return out

  @staticmethod
  def _set_requires_grad(opt_state, flag):
    for p in opt_state.params:
      p.requires_grad_(flag)

  def train_step(self, images, labels):
    """One unrolled step: images [(disc_iters+1)*B, H, W, C] fp32 in [0,1] and labels
    [(disc_iters+1)*B] int32, already on the device (modular_gan.py:428-469,568-584).

    Returns {"d_losses": [...], "g_loss": tensor} (device scalars)."""
    if not self._built:
      raise RuntimeError("call build() first")
    num_sub_steps = self._disc_iters + 1
    total = images.shape[0]
    if total % num_sub_steps:
      raise ValueError("batch of %d does not split into %d sub-steps" % (total, num_sub_steps))
    bsz = total // num_sub_steps
    fs, ls = [], []
    for i in range(num_sub_steps):
      f, l = self._preprocess(images[i * bsz:(i + 1) * bsz], labels[i * bsz:(i + 1) * bsz], i)
      fs.append(f)
      ls.append(l)
    d_losses = []
    # sub_step_hook (parity tests only, never under graph capture): called with ("d", i) before
    # discriminator sub-step i and ("g", disc_iters) before the generator sub-step, every update
    # enqueued so far joined -- tests take over the complete state there and compare each sub-step
    # from identical states (tests/gan_util.py stepwise_parity: the free-running comparison of a
    # whole step is chaotic, profiles/r05_gloss_spread.txt)
    hook = getattr(self, "sub_step_hook", None)
    with ops.use_store(self.store):
      self._generate_for_disc(fs)
      for i in range(self._disc_iters):
        if hook is not None:
          self._join_updates()
          hook("d", i)
        d_losses.append(self._disc_sub_step(fs[i], ls[i]))
      if hook is not None:
        self._join_updates()
        hook("g", self._disc_iters)
      g_loss = self._train_generator(fs[-1], ls[-1])
    self._join_updates()
    return {"d_losses": d_losses, "g_loss": g_loss}

  def unroll_graph(self, use_tpu=True):
    """modular_gan.py:533: the step is unrolled on accelerators or when forced by gin."""
    return bool(self._experimental_force_graph_unroll or use_tpu)

  def train_step_not_unrolled(self, images, labels):
    """One session.run of the reference's NOT unrolled graph (modular_gan.py:533-584 with
    use_tpu=False and experimental_force_graph_unroll=False): ONE sub-batch
    [B,H,W,C] -- one generator forward on it, one D update, and, when the discriminator step
    counter AFTER its increment is a multiple of disc_iters, one G update on the SAME z / images:
    the generator forward is the one built at the start of the step, the discriminator forward is
    fresh and sees the just-updated D (tf.control_dependencies, :573-576).  Otherwise g_loss = 0.

    Kept for parity with GPU runs of the reference; the accelerator path (and bench.py) is the
    unrolled train_step().  Reads the step counter on the host (one synchronisation)."""
    if not self._built:
      raise RuntimeError("call build() first")
    if self._experimental_joint_gen_for_disc:
      raise ValueError("Joining G forward passes is only supported for unrolled graphs.")
    # the random streams are keyed by (name, global_step): the calls between two G updates get
    # distinct names, as the reference's stateful random ops give them distinct draws
    self._join_updates()
    f, l = self._preprocess(images, labels, int(self.global_step_disc.item()) % self._disc_iters)
    with ops.use_store(self.store):
      sampled_y = None
      if self.conditional:
        sampled_y = self._get_one_hot_labels(f["sampled_labels"])
      self._set_requires_grad(self.g_opt, True)
      self.g_opt.join()
      f["generated"] = self.generator(f["z"], y=sampled_y, is_training=True)
      d_loss = self._train_discriminator(f, l)
      # the counter is incremented on the stream the update ran on: under data-parallel overlap
      # that is the communication stream, and .item() only synchronises the current one -- a stale
      # value would make ranks disagree on whether the G update (and its all-reduce) happens
      self.d_opt.join()
      if int(self.global_step_disc.item()) % self._disc_iters == 0:
        g_loss = self._train_generator(f, l, shared_forward=True)
      else:
        g_loss = torch.zeros((), dtype=torch.float32, device=self.device)
    self._join_updates()
    return {"d_losses": [d_loss], "g_loss": g_loss}

  def _join_updates(self):
    self.d_opt.join()
    self.g_opt.join()

  def _generate_for_disc(self, fs):
    """The generator forwards of all discriminator sub-steps as ONE batched call -- none of them
    sees a generator update (modular_gan.py:444-467 builds them all before the first D update).

    experimental_joint_gen_for_disc = True is the reference's own option (modular_gan.py:444-458):
    batch norm statistics over the joint batch.  Otherwise the calls are batched only when that
    is the same arithmetic as the separate calls: batch norm keeps one set of statistics per
    sub-step (ops.statistics_groups), and the generator must carry no per-call state besides the
    batch-norm moving averages -- a spectrally normalised generator runs one power iteration per
    call (arch_ops.py:479-535), so it keeps its separate calls."""
    n = self._disc_iters
    groups = self.joint_generation_groups()
    if groups is None:
      return
    joint = groups == 1
    with torch.no_grad():
      z = torch.cat([f["z"] for f in fs[:n]], dim=0)
      sampled_y = None
      if self.conditional:
        sampled_y = self._get_one_hot_labels(
            torch.cat([f["sampled_labels"] for f in fs[:n]], dim=0))
      self.g_opt.join()
      with ops.statistics_groups(1 if joint else n):
        generated = self.generator(z, y=sampled_y, is_training=True)
    bsz = z.shape[0] // n
    for i in range(n):
      fs[i]["generated"] = generated[i * bsz:(i + 1) * bsz]

  def joint_generation_groups(self):
    """How train_step() runs the generator forwards of the discriminator sub-steps: None =
    one call per sub-step; 1 = one joint call with joint batch-norm statistics (the reference's
    experimental_joint_gen_for_disc); disc_iters = one batched call with one set of statistics
    per sub-step (same arithmetic as separate calls, only for generators without per-call state)."""
    if self._disc_iters < 2:
      return None
    if self._experimental_joint_gen_for_disc:
      return 1
    if not _JOINT_G or self.store.sn_registry.get(self.generator.name):
      return None
    return self._disc_iters

  def _disc_sub_step(self, features, labels):
    """G forward (no gradient) on the sub-step's z + one D update (modular_gan.py:465-485)."""
    if "generated" not in features:
      with torch.no_grad():
        sampled_y = None
        if self.conditional:
          sampled_y = self._get_one_hot_labels(features["sampled_labels"])
        self.g_opt.join()
        features["generated"] = self.generator(features["z"], y=sampled_y, is_training=True)
    return self._train_discriminator(features, labels)

  def disc_step(self, images, labels):
    """ONE discriminator sub-step on its own (the unit BASELINE.json's north star quotes the
    128x128 ResNet on): fresh z, G forward, D forward + backward (+ penalty), D Adam update.
    images [B,H,W,C] fp32, labels [B] int32.  Returns {"d_loss": device scalar}."""
    if not self._built:
      raise RuntimeError("call build() first")
    f, l = self._preprocess(images, labels, 0)
    with ops.use_store(self.store):
      out = {"d_loss": self._disc_sub_step(f, l)}
    self._join_updates()
    return out

  # -- hipGraph capture of the whole step ---------------------------------------------------------------
  def capture_train_step(self, num_warmup=2):
    """Captures one unrolled step (all kernels of disc_iters D updates + the G update) into a
    hipGraph: small configs are launch-bound, replay removes the per-launch
    host cost.  Inputs are copied into static device buffers before each replay.  Returns
    run(images, labels) -> same dict as train_step."""
    return self._capture(self.train_step, self._disc_iters + 1, self._disc_iters, 1, num_warmup)

  def capture_disc_step(self, num_warmup=2):
    """disc_step() as a hipGraph: run(images [B,...], labels [B]) -> {"d_loss": ...}."""
    return self._capture(self.disc_step, 1, 1, 0, num_warmup)

  def _capture(self, step, nsub, d_updates, g_updates, num_warmup):
    # weight gradients become a parallel branch of the graph (CGAMD_NO_WGRAD_STREAM=1: A/B switch)
    Fn.enable_wgrad_stream(os.environ.get("CGAMD_WGRAD_STREAM", "") == "1")
    shape = (nsub * self.batch_size,) + tuple(self._dataset.image_shape)
    self._static_images = torch.zeros(shape, dtype=torch.float32, device=self.device)
    self._static_labels = torch.zeros((shape[0],), dtype=torch.int32, device=self.device)

    # the warm-up steps run for real (on the zero-filled static inputs): everything they touch --
    # variables, Adam slots, EMA shadows, step counters -- is put back afterwards, in place, so
    # that capturing leaves the training state exactly as it found it
    snapshot = self.state_dict()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(num_warmup):
        step(self._static_images, self._static_labels)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    self.d_opt.reserve_tables(d_updates)
    self.g_opt.reserve_tables(g_updates)
    if tpu_ops.data_parallel() and tpu_ops.thread_state() is None:
      # The process group's watchdog thread retires the warm-up's eager collectives by querying
      # their end events once per poll period (ProcessGroupNCCL: kWatchdogThreadSleepMillis =
      # 100 ms).  Once the captured collectives below pull RCCL's internal stream into the capture,
      # such a query fails with hipErrorCapturedEvent and the watchdog aborts the process.  After
      # the device synchronize above every one of those collectives HAS completed, so waiting a
      # little more than two poll periods is a bound, not a guess: the watchdog has seen them all.
      import time
      time.sleep(2.5 * _NCCL_WATCHDOG_POLL_S)
    self._graph = torch.cuda.CUDAGraph()
    graph_kwargs = {}
    if tpu_ops.num_replicas() > 1 or tpu_ops.force_data_parallel():
      # the process group's watchdog thread queries events while this thread captures: in the
      # default "global" capture mode such a call from ANOTHER thread invalidates the capture
      # (and aborts the process); thread-local mode confines the restrictions to this thread
      graph_kwargs["capture_error_mode"] = "thread_local"
    with torch.cuda.graph(self._graph, **graph_kwargs):
      self._graph_out = step(self._static_images, self._static_labels)
    for opt in (self.g_opt, self.d_opt):
      for t in opt.captured_tables:
        t.flush()
    self.load_state_dict(snapshot)
    del snapshot
    torch.cuda.synchronize()

    def run(images, labels):
      self._static_images.copy_(images, non_blocking=True)
      self._static_labels.copy_(labels, non_blocking=True)
      self._graph.replay()
      return self._graph_out

    return run

  # -- inference -----------------------------------------------------------------------------------
  def generate(self, z, labels=None, use_ema=None):
    """G(z, y) with is_training=False; EMA weights when g_use_ema (modular_gan.py:266-285)."""
    use_ema = self._g_use_ema if use_ema is None else use_ema
    swapped = None
    if use_ema and self.g_opt.ema is not None:
      swapped = [p.detach().clone() for p in self.g_opt.params]
      with torch.no_grad():
        for p, e in zip(self.g_opt.params, self.g_opt.ema):
          p.copy_(e)
    try:
      with torch.no_grad(), ops.use_store(self.store):
        y = self._get_one_hot_labels(labels) if self.conditional else None
        return self.generator(z, y=y, is_training=False)
    finally:
      if swapped is not None:
        with torch.no_grad():
          for p, s in zip(self.g_opt.params, swapped):
            p.copy_(s)

  def make_sampler(self, batch_size, use_ema=None):
    """sample(z [B, z_dim], labels [B] or None) -> the same images as generate(z, labels), as ONE
    hipGraph replay per batch: the evaluation samples its 10,000 images in batches of 64
    (eval_gan_lib.py:95-140), ~60 launches of a few microseconds each per batch when issued one by
    one.  The graph reads the variables (or their EMA shadows) where they live, so later in-place
    updates are seen; the result is a STATIC buffer, overwritten by the next call -- consume (or
    copy) it first, stream order is enough.  CGAMD_EVAL_GRAPH=0 (or no GPU) returns generate itself.
    The batch-norm accumulator fill (eval_gan_lib._update_bn_accumulators) changes host-side control
    flow and keeps calling generate()."""
    if not (torch.cuda.is_available() and self.device.type == "cuda") or \
        os.environ.get("CGAMD_EVAL_GRAPH", "1") == "0":
      return lambda z, labels=None: self.generate(z, labels, use_ema=use_ema)
    bsz = int(batch_size)
    z_static = torch.zeros((bsz, self._z_dim), dtype=torch.float32, device=self.device)
    y_static = None
    if self.conditional:
      y_static = torch.zeros((bsz,), dtype=torch.int32, device=self.device)
    side = torch.cuda.Stream(device=self.device)
    side.wait_stream(torch.cuda.current_stream(self.device))
    with torch.cuda.stream(side):          # first use of every kernel / workspace outside the capture
      self.generate(z_static, y_static, use_ema=use_ema)
    torch.cuda.current_stream(self.device).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    kwargs = {}
    if tpu_ops.num_replicas() > 1 or tpu_ops.force_data_parallel():
      kwargs["capture_error_mode"] = "thread_local"    # see _capture: the watchdog thread
    with torch.cuda.graph(graph, **kwargs):
      out = self.generate(z_static, y_static, use_ema=use_ema)

    def sample(z, labels=None):
      z_static.copy_(z, non_blocking=True)
      if y_static is not None:
        y_static.copy_(labels, non_blocking=True)
      graph.replay()
      return out

    sample.graph = graph
    return sample

  # -- checkpoint state (SURVEY section 5 / App. D naming) -------------------------------------------------
  def state_dict(self):
    sd = dict(self.store.state_dict())
    for opt, tag in ((self.g_opt, "g_opt"), (self.d_opt, "d_opt")):
      for n, m, v in zip(opt.names, opt.m, opt.v):
        sd["%s/%s/Adam" % (n, tag)] = m
        sd["%s/%s/Adam_1" % (n, tag)] = v
      if opt.ema is not None:
        for n, e in zip(opt.names, opt.ema):
          sd[n + "/ExponentialMovingAverage"] = e
    sd["global_step"] = self.global_step
    sd["global_step_disc"] = self.global_step_disc
    return {k: v.detach().clone() for k, v in sd.items()}

  def load_state_dict(self, sd):
    with torch.no_grad():
      for name, var in self.store.vars.items():
        var.copy_(sd[name])
      for opt, tag in ((self.g_opt, "g_opt"), (self.d_opt, "d_opt")):
        for n, m, v in zip(opt.names, opt.m, opt.v):
          m.copy_(sd["%s/%s/Adam" % (n, tag)])
          v.copy_(sd["%s/%s/Adam_1" % (n, tag)])
        if opt.ema is not None:
          for n, e in zip(opt.names, opt.ema):
            e.copy_(sd[n + "/ExponentialMovingAverage"])
      self.global_step.copy_(sd["global_step"])
      self.global_step_disc.copy_(sd["global_step_disc"])
