"""Import-path shim: put ``gnn_matlang_amd`` on sys.path ahead of the reference checkout and
``from libs.spect_conv import SpectConv, ML3Layer`` / ``from libs.utils import SpectralDesign``
resolve to the MI355X implementation."""
