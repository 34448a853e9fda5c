"""Regenerates the fixtures under tests/golden/ from the reference checkout (run in the build
container, where /root/reference exists; the GPU box only ever sees the committed files).

  * example_configs/*.gin : the reference's five runnable configs, byte-for-byte -- the drop-in
    contract is that they parse and bind unchanged.
  * reference_pins.json   : the golden numbers the reference's own tests assert, with the
    file:line they come from.
"""
import json
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _variable_lists(path):
  """The (name, shape) literals of every expected_variables list in a reference test file."""
  src = open(path).read()
  parts = re.split(r"\n  def (test\w+)\(self\):", src)
  out = {}
  for i in range(1, len(parts), 2):
    items = re.findall(r'\("([^"]+):0",\s*\[([0-9,* ]*)\]\)', parts[i + 1])
    if items:
      out[parts[i]] = [[n, [int(eval(d)) for d in dims.split(",") if d.strip()]]  # "3 * 3 * 256"
                       for n, dims in items]
  return out


def main():
  dst = os.path.join(HERE, "example_configs")
  os.makedirs(dst, exist_ok=True)
  for name in sorted(os.listdir(os.path.join(REF, "example_configs"))):
    if name.endswith(".gin"):
      shutil.copyfile(os.path.join(REF, "example_configs", name), os.path.join(dst, name))
  pins = {
      "batch_norm_golden": {
          "source": "compare_gan/architectures/arch_ops_test.py:32-61",
          "epsilon": 1e-3,
          "x": [[[[5, 7, 2]], [[5, 8, 8]]], [[[1, 2, 0]], [[4, 0, 4]]],
                [[[6, 2, 6]], [[5, 0, 5]]], [[[2, 4, 2]], [[6, 4, 1]]]],
          "expected": [[[[0.4375205, 1.30336881, -0.58830315]], [[0.4375205, 1.66291881, 1.76490951]]],
                       [[[-1.89592218, -0.49438119, -1.37270737]], [[-0.14584017, -1.21348119, 0.19610107]]],
                       [[[1.02088118, -0.49438119, 0.98050523]], [[0.4375205, -1.21348119, 0.58830321]]],
                       [[[-1.31256151, 0.22471881, -0.58830315]], [[1.02088118, 0.22471881, -0.98050523]]]],
      },
      "fid_golden": {"source": "compare_gan/metrics/fid_score_test.py:31-40", "value": 89.091,
                     "tolerance": 1e-4},
      "param_counts": {
          "source": "compare_gan/architectures/resnet_biggan_test.py:139,154; "
                    "resnet_biggan.py:39-62; resnet_norm_test.py:124-162; "
                    "resnet_biggan_deep_test.py:31-60 (z_dim 128, 1000 classes, 128 px, "
                    "conditional batch norm, default ch = 128)",
          "resnet_biggan_arch_128": {"G": 70433988, "D": 87982370},
          "resnet_biggan_deep_arch_128": {"G": 50244484, "D": 34590210},
          "resnet_cifar_arch": {"G": 5849603, "D": 1483137},
      },
      "resnet_cifar_variables": dict(
          source="compare_gan/architectures/resnet_norm_test.py:30-369 (expected_variables lists; "
                 "the last test lists tf.global_variables, the others tf.trainable_variables)",
          **_variable_lists(os.path.join(REF, "compare_gan/architectures/resnet_norm_test.py"))),
      "step_counters": {"source": "compare_gan/gans/modular_gan_test.py:175-177",
                        "rule": "global_step_disc == steps * disc_iters; global_step == steps"},
  }
  with open(os.path.join(HERE, "reference_pins.json"), "w") as f:
    json.dump(pins, f, indent=1)


if __name__ == "__main__":
  main()
