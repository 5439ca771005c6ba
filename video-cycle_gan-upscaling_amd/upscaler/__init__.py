"""MI355X-native drop-in for the reference's ``upscaler`` package (GAN hot path only).

    from upscaler.model import make_upscaler_orig, make_discriminator_simple_512, make_and_compile_gan2
    from upscaler.data import convert_image_series_to_array, convert_array_to_image

mirrors the import block of upscaling/train_gan3.py:1-8.  See DESIGN.md for scope.
"""
