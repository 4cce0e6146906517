"""Identity quantizer (reference: pytorch_quantizer/quantization/qtypes/dummy_quantizer.py:2-7),
used for fp32 weights (-qw f32) and biases."""


class DummyQuantizer:
    def __call__(self, tensor, tag="", stat_id=None, override_att=None):
        return tensor

    def __repr__(self):
        return 'DummyQuantizer - fp32'
