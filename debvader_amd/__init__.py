"""debvader_amd — MI355X-native engine behind debvader's create_model_vae / train / deblend() surface.

Python host code over a C-ABI HIP library (debvader_amd/lib/libdebvader_hip.so, include/debvader_hip.h).
"""
__version__ = "0.1.0"
