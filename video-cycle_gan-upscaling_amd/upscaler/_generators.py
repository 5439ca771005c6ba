"""The other generator topologies train_gan3.py offers behind ``-gm`` (upscaling/train_gan3.py:55,234-252), written block by block
as the reference writes them, on the functional graph API of ``_graph`` (same layer names -- Keras' automatic ones where the
reference names nothing -- same weights, same kernels):

  make_upscaler_skip_con          upscaling/upscaler/model.py:332-363   ('skip-con')
  make_upscaler_unetish           model.py:570-634, blocks :505-566      ('unetish')
  make_upscaler_unetish_add       model.py:642-716                       ('unetish-add')
  make_upscaler_unetish_complex   model.py:743-827                       (defined, offered by no driver)

  make_upscaler_incep_resnet      model.py:443-497, blocks :372-440      ('inc-resnet': 1x1 / 1xk / kx1 convolutions on 19-, 25-, 32-, 48-channel paths)
"""
import math

from . import _graph as G


# ---- U-Net-ish blocks (model.py:505-566) ----------------------------------------------------------------------------------------------
def same_size_unetish_block(model, kernel_size, filters, strides, name, dropout_rate=0.1):
    """model.py:505-512: Conv2D -> BatchNormalization() (unnamed) -> PReLU -> Dropout"""
    model = G.conv2d(model, filters, kernel_size, strides, "same", name=name + "/Conv2D")
    model = G.batch_norm_prelu(model, None, name + "/PReLU")
    return G.dropout(model, dropout_rate, name=name + "/Dropout")


downsampling_unetish_block = same_size_unetish_block          # model.py:514-521: the same four layers, called with strides=2


def upsampling_unetish_block(model, kernel_size, filters, strides, name, dropout_rate=0.1):
    """model.py:523-530: Conv2DTranspose -> BatchNormalization() -> PReLU -> Dropout"""
    model = G.conv2d_transpose(model, filters, kernel_size, strides, name=name + "/Conv2DTrans")
    model = G.batch_norm_prelu(model, None, name + "/PReLU")
    return G.dropout(model, dropout_rate, name=name + "/Dropout")


def find_crop_shape(output_down, output_up):
    """model.py:533-546 (the reference reads the two shapes off throw-away Models; the graph carries them)"""
    g = output_down.graph
    (dh, dw), (uh, uw) = g.hw[output_down.nid], g.hw[output_up.nid]
    height_diff, width_diff = uh - dh, uw - dw
    top_crop, left_crop = height_diff // 2, width_diff // 2
    return ((top_crop, height_diff - top_crop), (left_crop, width_diff - left_crop))


def concatenate_layers(input_layer, output_down, output_up, name):
    """model.py:549-556"""
    model = G.cropping2d(output_up, find_crop_shape(output_down, output_up), name=name + "/Cropping2D")
    return G.concatenate([output_down, model], name=name + "/Concatenate")


def sum_layers(input_layer, output_down, output_up, name):
    """model.py:559-566"""
    model = G.cropping2d(output_up, find_crop_shape(output_down, output_up), name=name + "/Cropping2D")
    return G.add([output_down, model], name=name + "/Add")


def _final_crop(model, output_image_shape, name):
    """model.py:613-626: crop whatever the U produced beyond the requested output"""
    h, w = model.graph.hw[model.nid]
    height_diff, width_diff = h - output_image_shape[0], w - output_image_shape[1]
    top_crop, left_crop = height_diff // 2, width_diff // 2
    return G.cropping2d(model, ((top_crop, height_diff - top_crop), (left_crop, width_diff - left_crop)), name=name)


def _u(upscaler_input, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, join, halve_after_bottom):
    """the U shared by the three variants (model.py:577-609, 649-685, 750-786)"""
    upscale_times = int(math.log(upscale_factor, 2)) + downscale_times
    model = G.conv2d(upscaler_input, initial_step_filter_count, 9, 1, "same", name="initial/Conv2D")
    model = G.prelu(model, name="initial/PReLU")
    outputs = []
    step_filter_count = initial_step_filter_count
    step = 0
    for step in range(downscale_times):
        for index in range(step_size):
            # NB the reference does not pass dropout_rate here: these blocks keep the default 0.1 whatever -dr says (model.py:590)
            model = same_size_unetish_block(model, kernel_size, step_filter_count, 1, "down/" + str(step) + "/same/" + str(index))
        outputs.append(model)
        model = downsampling_unetish_block(model, kernel_size, step_filter_count, 2, "down/" + str(step) + "/down", dropout_rate=dropout_rate)
        step_filter_count = step_filter_count * 2
    for index in range(step_size):
        model = same_size_unetish_block(model, kernel_size, step_filter_count, 1, "bottom/" + str(step) + "/same/" + str(index), dropout_rate=dropout_rate)
    if halve_after_bottom:
        step_filter_count = step_filter_count // 2
    down_outputs_len = len(outputs)
    for step in range(upscale_times):
        model = upsampling_unetish_block(model, kernel_size, step_filter_count, 2, "up/" + str(step) + "/up", dropout_rate=dropout_rate)
        if step < down_outputs_len:
            model = join(upscaler_input, outputs[down_outputs_len - step - 1], model, "up/" + str(step) + ("/add" if join is sum_layers else "/concat"))
            step_filter_count = step_filter_count // 2
        for index in range(step_size):
            model = same_size_unetish_block(model, kernel_size, step_filter_count, 1, "up/" + str(step) + "/same/" + str(index), dropout_rate=dropout_rate)
    return model


def _input(output_image_shape, upscale_factor):
    shape = (output_image_shape[0] // upscale_factor, output_image_shape[1] // upscale_factor, output_image_shape[2])
    return G.Input(shape=shape, name="input")


def make_upscaler_unetish(output_image_shape, kernel_size=5, upscale_factor=4, step_size=4, downscale_times=5, initial_step_filter_count=32,
                          dropout_rate=0.1, seed=7):
    """model.py:570-634"""
    upscaler_input = _input(output_image_shape, upscale_factor)
    model = _u(upscaler_input, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, concatenate_layers, False)
    model = G.conv2d(model, 3, 9, 1, "same", activation="tanh")                                   # unnamed: conv2d_1 (+ activation_1)
    model = _final_crop(model, output_image_shape, "final/Cropping2D")
    return G.build_model(upscaler_input, model, name="upscaler_unetish", seed=seed)


def make_upscaler_unetish_add(output_image_shape, kernel_size=5, upscale_factor=4, step_size=4, downscale_times=5, initial_step_filter_count=48,
                              dropout_rate=0.1, seed=7):
    """model.py:642-716: the joins are sums, and the bilinearly resized, atanh-mapped input is added before one more 9x9 convolution"""
    upscaler_input = _input(output_image_shape, upscale_factor)
    model = _u(upscaler_input, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, sum_layers, True)
    model = G.conv2d(model, 3, 9, 1, "same", activation="tanh")                                   # conv2d_1
    model = _final_crop(model, output_image_shape, "prefinal/Cropping2D")
    resized_input = G.resize_images(upscaler_input, upscale_factor, "bilinear", name="final/input_resize/resize")
    resized_input = G.atanh_scaled(resized_input, 0.99999, name="final/input_resize/atanh")
    model = sum_layers(upscaler_input, model, resized_input, "final/concat")
    model = G.conv2d(model, 3, 9, 1, "same", activation="tanh")                                   # conv2d_2
    return G.build_model(upscaler_input, model, name="upscaler_unetish_add", seed=seed)


def make_upscaler_unetish_complex(output_image_shape, kernel_size=5, upscale_factor=4, step_size=4, downscale_times=3, initial_step_filter_count=32,
                                  dropout_rate=0.1, seed=7):
    """model.py:743-827: the U, then three rounds of a 3-channel attention head driven by the resized input"""
    upscaler_input = _input(output_image_shape, upscale_factor)
    model = _u(upscaler_input, kernel_size, upscale_factor, step_size, downscale_times, initial_step_filter_count, dropout_rate, concatenate_layers, False)
    resized_input = G.resize_images(upscaler_input, upscale_factor, "bilinear", name="input_resize/resize")
    attention = G.conv2d(resized_input, 3, 9, 1, "same", name="final/initial/attention")
    for step in range(3):
        p = "final/" + str(step)
        attention = G.concatenate([resized_input, attention], name=p + "/input_concat")
        pre = G.conv2d(attention, 3, 9, 1, "same", name=p + "/attention")
        attention = G.activation(pre, "sigmoid", name=p + "/att_sigmoid")                        # feeds the next round's concatenation (:799)
        model = G.conv2d(model, 3, 9, 1, "same", name=p + "/Conv2D")
        att_model = G.multiply_sigmoid(pre, model, name=p + "/att_Conv2D")                       # Multiply([sigmoid(pre), model]) (:802), fused
        model = G.concatenate([att_model, model], name=p + "/input_att_concat")
        model = G.conv2d(model, 3, 9, 1, "same", activation="tanh", name=p + "/Conv2D_after_att")
        if step < 2:
            model = G.dropout(model, dropout_rate, name=p + "/Dropout")
    model = _final_crop(model, output_image_shape, "final/Cropping2D")
    return G.build_model(upscaler_input, model, name="upscaler_unetish_complex", seed=seed)


def make_upscaler_skip_con(output_image_shape, kernel_size=5, filters=64, upscale_factor=4, unique_names=False, seed=7):
    """model.py:332-363: make_upscaler_orig with 16 residual blocks, 224-filter up-sampling blocks and the bilinearly resized input
    concatenated in front of the last convolution.

    As written the reference cannot build this model: it calls ``residual_block(model, kernel_size, filters, 1)`` sixteen times
    without a name, so sixteen layers are called '/conv_pre' (...) and keras.engine.network refuses duplicate layer names.  The same
    happens here (ValueError) unless ``unique_names=True``, which numbers the blocks the way make_upscaler_orig does."""
    input_image_shape = (output_image_shape[0] // upscale_factor, output_image_shape[1] // upscale_factor, output_image_shape[2])
    upscale_times = int(math.log(upscale_factor, 2))
    upscaler_input = G.Input(shape=input_image_shape)
    model = G.conv2d(upscaler_input, 64, 9, 1, "same")
    model = G.prelu(model)
    upsc_model = model
    for index in range(16):
        model = G.residual_block(model, kernel_size, filters, 1, name="res_block/" + str(index) if unique_names else "")
    model = G.conv2d(model, 64, 3, 1, "same")
    model = G.batch_norm(model)
    model = G.add([upsc_model, model])
    for index in range(upscale_times):
        model = G.upsampling_block(model, 3, 224, 2, name="upscaling/" + str(index) + "/block" if unique_names else "")
    resized_input = G.resize_images(upscaler_input, 2 ** upscale_times, "bilinear")
    model = G.concatenate([resized_input, model])
    model = G.conv2d(model, 3, 9, 1, "same", activation="tanh")
    return G.build_model(upscaler_input, model, name="upscaler_skip_con", seed=seed)


# ---- inception-resnet (model.py:372-497) -----------------------------------------------------------------------------------------------
def inception_mini_resblock(model, filters, name, kernel_size, batch_normalisation=True):
    """model.py:372-382: [BatchNormalization] -> PReLU -> Conv2D(kernel_size = (kh, kw)), pre-activation order"""
    if batch_normalisation:
        model = G.batch_norm_prelu(model, name + "/batch_norm", name + "/prelu")
    else:
        model = G.prelu(model, name=name + "/prelu")
    return G.conv2d(model, filters, tuple(kernel_size), 1, "same", name=name + "/%dx%d" % (kernel_size[0], kernel_size[1]))


def inception_resblock_3path(model, filters, name, kernel_size=3, batch_normalisation=True):
    """model.py:386-412"""
    gen = model
    path_a_filters = int(filters * 0.5)
    path_b_filters = int(filters * 0.5)
    path_c_filters1 = int(filters * 0.5)
    path_c_filters2 = int(filters * 0.75)
    path_c_filters3 = filters
    bn, k = batch_normalisation, kernel_size
    path_a_model = inception_mini_resblock(model, path_a_filters, name + "/a/1", (1, 1), bn)
    path_b_model = inception_mini_resblock(model, path_b_filters, name + "/b/1", (1, 1), bn)
    path_b_model = inception_mini_resblock(path_b_model, path_b_filters, name + "/b/2", (k, k), bn)
    path_c_model = inception_mini_resblock(model, path_c_filters1, name + "/c/1", (1, 1), bn)
    path_c_model = inception_mini_resblock(path_c_model, path_c_filters2, name + "/c/2", (k, k), bn)
    path_c_model = inception_mini_resblock(path_c_model, path_c_filters3, name + "/c/3", (k, k), bn)
    model = G.concatenate([path_a_model, path_b_model, path_c_model], name=name + "/final/concat")
    model = G.conv2d(model, filters, 1, 1, "same", name=name + "/final/1x1")
    return G.add([gen, model], name=name + "/final/add")


def inception_resblock_2path(model, filters, name, kernel_size=7, batch_normalisation=True):
    """model.py:416-439"""
    gen = model
    path_a_filters = int(filters * 0.5)
    path_b_filters1 = int(filters * 0.3)
    path_b_filters2 = int(filters * 0.4)
    path_b_filters3 = int(filters * 0.5)
    bn, k = batch_normalisation, kernel_size
    path_a_model = inception_mini_resblock(model, path_a_filters, name + "/a/1", (1, 1), bn)
    path_b_model = inception_mini_resblock(model, path_b_filters1, name + "/b/1", (1, 1), bn)
    path_b_model = inception_mini_resblock(path_b_model, path_b_filters2, name + "/b/2", (1, k), bn)
    path_b_model = inception_mini_resblock(path_b_model, path_b_filters3, name + "/b/3", (k, 1), bn)
    model = G.concatenate([path_a_model, path_b_model], name=name + "/final/concat")
    model = G.conv2d(model, filters, 1, 1, "same", name=name + "/final/1x1")
    return G.add([gen, model], name=name + "/final/add")


def make_upscaler_incep_resnet(output_image_shape, filters=64, upscale_factor=4,
                               a_block_type="3path", a_block_num=5, a_block_kernel=3,
                               b_block_type="2path", b_block_num=10, b_block_kernel=7,
                               c_block_type="2path", c_block_num=5, c_block_kernel=3, seed=7):
    """model.py:443-497.  Kernels instantiated: 1x1, 3x3 / 5x5 (3-path blocks), 1xk / kx1 with k in 3, 5, 7 (2-path blocks)."""
    input_image_shape = (output_image_shape[0] // upscale_factor, output_image_shape[1] // upscale_factor, output_image_shape[2])
    upscale_times = int(math.log(upscale_factor, 2))
    upscaler_input = G.Input(shape=input_image_shape, name="initial/input")
    model = G.conv2d(upscaler_input, filters, 9, 1, "same", name="initial/conv/9x9")
    upsc_model = model
    for tag, btype, num, kern in (("A", a_block_type, a_block_num, a_block_kernel), ("B", b_block_type, b_block_num, b_block_kernel),
                                  ("c", c_block_type, c_block_num, c_block_kernel)):            # 'c' in lower case: model.py:479,481
        for index in range(num):
            if btype == "3path":
                model = inception_resblock_3path(model, filters, "inc_res_block/%s/3p/%d" % (tag, index), kernel_size=kern, batch_normalisation=True)
            elif btype == "2path":
                model = inception_resblock_2path(model, filters, "inc_res_block/%s/2p/%d" % (tag, index), kernel_size=kern, batch_normalisation=True)
    model = G.conv2d(model, filters, c_block_kernel, 1, "same", name="prefinal/conv2d")
    model = G.batch_norm(model, name="prefinal/batch_norm")
    model = G.add([upsc_model, model], name="prefinal/tanh")
    for index in range(upscale_times):
        model = G.upsampling_block(model, c_block_kernel, 256, 2, name="upscaling/" + str(index) + "/block")
    model = G.conv2d(model, 3, 9, 1, "same", activation="tanh", name="final/conv")
    return G.build_model(upscaler_input, model, name="upscaler_incep_resnet", seed=seed)
