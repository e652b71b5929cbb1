from ..spectral_design import SpectralDesign  # noqa: F401


def get_n_params(model):
    """number of scalar parameters (reference: libs/utils.py:14-21)."""
    return sum(p.numel() for p in model.parameters())
