"""compare_gan_amd/graphdef.py: the frozen-GraphDef reader behind `inception_weights.path = *.pb`
(eval_utils.py:41-49 of the reference reads inceptionv1_for_inception_score.pb).  The real file is
not available offline, so a SYNTHETIC graph with the 2015 graph's node naming and batch-norm
structure is written in GraphDef wire format, read back, folded, and compared with the folding done
by hand; google.protobuf (installed, without TensorFlow's message classes) cross-checks the wire
format through its generic decoder where it can."""
import numpy as np

from compare_gan_amd import graphdef
from compare_gan_amd import inception


def _synthetic_graph(rng, scale_after=False):
    consts, others, want = {}, [], {}
    shapes, c_final = inception.conv_shapes()
    assert c_final == inception.POOL3_DIM and len(shapes) == len(graphdef.GRAPH_SCOPE) == 94
    for name, (kh, kw, ci, co) in shapes:
        scope = graphdef.GRAPH_SCOPE[name]
        w = rng.standard_normal((kh, kw, ci, co)).astype(np.float32) * 0.05
        beta = rng.standard_normal(co).astype(np.float32) * 0.1
        gamma = (1.0 + 0.2 * rng.standard_normal(co)).astype(np.float32)
        mean = rng.standard_normal(co).astype(np.float32) * 0.1
        var = (0.5 + rng.random(co)).astype(np.float32)
        consts[scope + "/conv2d_params"] = w
        consts[scope + "/batchnorm/beta"] = beta
        consts[scope + "/batchnorm/gamma"] = gamma
        consts[scope + "/batchnorm/moving_mean"] = mean
        consts[scope + "/batchnorm/moving_variance"] = var
        others.append((scope + "/batchnorm", "BatchNormWithGlobalNormalization",
                       [scope + "/Conv2D", scope + "/batchnorm/moving_mean"],
                       {"variance_epsilon": 1e-3}))
        inv = 1.0 / np.sqrt(var.astype(np.float64) + 1e-3)
        want[name + "/kernel"] = (w.astype(np.float64) * inv).astype(np.float32)
        want[name + "/bias"] = (beta.astype(np.float64) - mean.astype(np.float64) * inv).astype(np.float32)
    consts[graphdef.LOGITS_WEIGHTS] = rng.standard_normal((2048, 1008)).astype(np.float32) * 0.02
    consts[graphdef.LOGITS_BIASES] = rng.standard_normal(1008).astype(np.float32) * 0.01
    want["logits/kernel"] = consts[graphdef.LOGITS_WEIGHTS]
    want["logits/bias"] = consts[graphdef.LOGITS_BIASES]
    return consts, others, want


def test_graphdef_round_trip_and_batch_norm_folding(tmp_path):
    rng = np.random.default_rng(5)
    consts, others, want = _synthetic_graph(rng)
    path = str(tmp_path / "inception_synthetic.pb")
    graphdef.write_graphdef(path, consts, others)
    nodes = graphdef.read_graphdef(path)
    assert len(nodes) == len(consts) + len(others)
    for name, arr in consts.items():
        assert nodes[name]["op"] == "Const"
        np.testing.assert_array_equal(nodes[name]["attrs"]["value"], arr)
    bn = nodes["mixed_10/tower_1/mixed/conv_1/batchnorm"]
    assert bn["op"] == "BatchNormWithGlobalNormalization" and len(bn["inputs"]) == 2
    assert abs(bn["attrs"]["variance_epsilon"] - 1e-3) < 1e-9
    got = graphdef.inception_weights_from_graphdef(path)
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == np.float32 and got[k].shape == want[k].shape
        np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-7)
    # every key InceptionV3.load_weights() expects is produced, with the expected shapes
    ref = inception.make_weights(seed=1)
    assert set(ref) == set(got)
    for k, v in ref.items():
        assert tuple(v.shape) == got[k].shape, k


def test_graphdef_wire_format_matches_google_protobuf(tmp_path):
    """The bytes write_graphdef emits decode with protobuf's own runtime: a descriptor for the four
    messages is built on the fly (field numbers as in tensorflow/core/framework/*.proto)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="mini_graph.proto", package="mini", syntax="proto3")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".mini." + tname
        return m
    T = descriptor_pb2.FieldDescriptorProto
    msg("Dim", [("size", 1, T.TYPE_INT64, T.LABEL_OPTIONAL, None)])
    msg("Shape", [("dim", 2, T.TYPE_MESSAGE, T.LABEL_REPEATED, "Dim")])
    msg("Tensor", [("dtype", 1, T.TYPE_INT32, T.LABEL_OPTIONAL, None),
                   ("tensor_shape", 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, "Shape"),
                   ("tensor_content", 4, T.TYPE_BYTES, T.LABEL_OPTIONAL, None)])
    msg("AttrValue", [("f", 4, T.TYPE_FLOAT, T.LABEL_OPTIONAL, None),
                      ("type", 6, T.TYPE_INT32, T.LABEL_OPTIONAL, None),
                      ("tensor", 8, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, "Tensor")])
    msg("AttrEntry", [("key", 1, T.TYPE_STRING, T.LABEL_OPTIONAL, None),
                      ("value", 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, "AttrValue")])
    msg("Node", [("name", 1, T.TYPE_STRING, T.LABEL_OPTIONAL, None),
                 ("op", 2, T.TYPE_STRING, T.LABEL_OPTIONAL, None),
                 ("input", 3, T.TYPE_STRING, T.LABEL_REPEATED, None),
                 ("attr", 5, T.TYPE_MESSAGE, T.LABEL_REPEATED, "AttrEntry")])
    msg("Graph", [("node", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, "Node")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Graph = message_factory.GetMessageClass(pool.FindMessageTypeByName("mini.Graph"))
    rng = np.random.default_rng(9)
    consts = {"a/conv2d_params": rng.standard_normal((3, 3, 4, 5)).astype(np.float32),
              "softmax/biases": rng.standard_normal(7).astype(np.float32)}
    path = str(tmp_path / "tiny.pb")
    graphdef.write_graphdef(path, consts, [("a/batchnorm", "BatchNormWithGlobalNormalization",
                                           ["a/Conv2D"], {"variance_epsilon": 1e-3})])
    g = Graph()
    g.ParseFromString(open(path, "rb").read())
    assert [n.name for n in g.node] == ["a/conv2d_params", "softmax/biases", "a/batchnorm"]
    t = [e.value.tensor for e in g.node[0].attr if e.key == "value"][0]
    assert t.dtype == 1 and [d.size for d in t.tensor_shape.dim] == [3, 3, 4, 5]
    np.testing.assert_array_equal(np.frombuffer(t.tensor_content, "<f4").reshape(3, 3, 4, 5),
                                  consts["a/conv2d_params"])
    assert abs(g.node[2].attr[0].value.f - 1e-3) < 1e-9 and g.node[2].input[0] == "a/Conv2D"
    # and the other way round: protobuf's serialisation is read by read_graphdef
    open(path, "wb").write(g.SerializeToString())
    nodes = graphdef.read_graphdef(path)
    np.testing.assert_array_equal(nodes["softmax/biases"]["attrs"]["value"], consts["softmax/biases"])
