"""Identity quantizer (reference: pytorch_quantizer/quantization/qtypes/dummy_quantizer.py:2-7),
used for fp32 weights (-qw f32) and biases.  It takes the reference's four positional slots (the
manager's five-argument call would raise TypeError there as well) and accepts the
attributes the manager sets on every quantizer (half_range, ...)."""

_LABEL = 'DummyQuantizer - fp32'


class DummyQuantizer(object):
    def __repr__(self):
        return _LABEL

    def __call__(self, tensor, tag="", stat_id=None, override_att=None):
        # nothing to compute: fp32 passes through, no device work is enqueued
        del tag, stat_id, override_att
        return tensor
