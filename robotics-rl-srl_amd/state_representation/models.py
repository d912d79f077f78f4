"""SRL encoder forward pass for the raw_pixels path (state_representation/models.py:38-193,
rl_baselines/utils.py:162-191), batched and device-resident.

The reference runs ONE encoder process that serves every env serially at batch size 1 through
multiprocessing queues (150 KB pickled per observation).  Here the rasteriser's uint8 NHWC batch stays
in HBM and goes through one batched forward: the fused HIP kernel csrc/encoder.hip (split-float16 MFMA, whole
network per image inside one CU) for 64x64x3 frames, the PyTorch-ROCm forward (MIOpen / rocBLAS) for every other
shape — the PyTorch forward is also the float32 reference the kernel is tested against.

srl_zoo is an empty submodule in the reference checkout, so the architecture is restated from its
published description: CustomCNN = conv7x7/2(3->64)+BN+ReLU+maxpool3/2,
conv3x3(64->64)+BN+ReLU+maxpool3/2, conv3x3/2(64->64)+BN+ReLU+maxpool3/2, FC(flat -> state_dim); the
flattened size follows the input size (6*6*64 at 224x224, 1*1*64 at 64x64).  Weights are random-initialised
unless a state_dict is given: there are no checkpoints in this environment."""
import numpy as np
import torch as th
import torch.nn as nn

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess(images_u8):
    """srl_zoo preprocessImage + the reference's layout: uint8 [N][H][W][C] -> float32 [N][C][W][H]
    (state_representation/models.py:185-188 transposes (0, 3, 2, 1): width and height swapped), scaled to
    [0, 1] and ImageNet mean/std normalised per 3-channel group.  Runs on the tensor's device."""
    x = images_u8.to(th.float32) / 255.0
    c = x.shape[-1]
    mean = th.tensor(IMAGENET_MEAN * (c // 3), dtype=th.float32, device=x.device)
    std = th.tensor(IMAGENET_STD * (c // 3), dtype=th.float32, device=x.device)
    x = (x - mean) / std
    return x.permute(0, 3, 2, 1).contiguous()


class CustomCNN(nn.Module):
    def __init__(self, state_dim=2, n_channels=3, img_shape=(224, 224)):
        super(CustomCNN, self).__init__()
        self.conv_layers = nn.Sequential(
            nn.Conv2d(n_channels, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            nn.Conv2d(64, 64, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(64, 64, kernel_size=3, stride=2, padding=1, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
        )
        with th.no_grad():
            flat = self.conv_layers(th.zeros(1, n_channels, img_shape[1], img_shape[0])).numel()
        self.flat_dim = flat
        self.fc = nn.Linear(flat, state_dim)

    def forward(self, x):
        x = self.conv_layers(x)
        return self.fc(x.reshape(x.size(0), -1))

    def getStates(self, x):
        return self.forward(x)


def fold_batchnorm(seq):
    """Inference-time fusion of every Conv2d + BatchNorm2d(eval) pair into one Conv2d with bias
    (w' = w * gamma / sigma, b' = beta - mu * gamma / sigma): removes three memory-bound BN passes per forward."""
    out, layers = [], list(seq)
    i = 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, nn.Conv2d) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.BatchNorm2d):
            bn = layers[i + 1]
            scale = bn.weight / th.sqrt(bn.running_var + bn.eps)
            conv = nn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, bias=True)
            conv.weight.data = (m.weight * scale.view(-1, 1, 1, 1)).detach().clone()
            bias = m.bias if m.bias is not None else th.zeros_like(bn.running_mean)
            conv.bias.data = ((bias - bn.running_mean) * scale + bn.bias).detach().clone()
            out.append(conv)
            i += 2
        else:
            out.append(m)
            i += 1
    return nn.Sequential(*out)


class SRLNeuralNetwork(object):
    """state_representation/models.py:SRLNeuralNetwork — getState(observation) for one image (reference
    surface) and getStates(images) for a whole device-resident batch."""

    def __init__(self, state_dim, cuda=False, model_type="custom_cnn", n_channels=3, img_shape=(224, 224),
                 state_dict=None, device=None, backend="auto"):
        assert model_type == "custom_cnn", "only the srl_zoo CustomCNN encoder is restated"
        self.state_dim = state_dim
        self.device = th.device(device if device is not None else ("cuda" if cuda else "cpu"))
        self.model = CustomCNN(state_dim, n_channels, img_shape)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        self.model = self.model.eval()
        with th.no_grad():
            self.fused_conv = fold_batchnorm(self.model.conv_layers).eval().to(self.device)
        self.model = self.model.to(self.device)
        # NHWC activations on the GPU: MIOpen's gfx950 solvers for these shapes are ~25 % faster (3.3 vs 4.3 ms per
        # 4096x64x64 batch); logical shapes, and therefore the flatten order in front of the FC, do not change
        self.memory_format = th.channels_last if self.device.type == "cuda" else th.contiguous_format
        self.fused_conv = self.fused_conv.to(memory_format=self.memory_format)
        # HIP forward through the C-ABI: the fused single-kernel path for 64x64x3 frames (csrc/encoder.hip), the layered
        # split-f16 MFMA path for every other shape — 224x224x3, 6-channel multi_view (csrc/encoder_general.hip).
        # backend: "auto" | "hip" | "torch" (the PyTorch-ROCm forward above).
        self.hip = None
        self.backend = "torch"
        if backend not in ("auto", "hip", "torch"):
            raise ValueError("backend must be auto, hip or torch")
        if backend != "torch" and self.device.type == "cuda":
            from srlhip import _lib
            if _lib.encoder_supported(img_shape[0], img_shape[1], n_channels):
                self.hip = _lib.Encoder(self.device.index or 0, img_shape, n_channels, state_dim, *self.folded_weights())
                self.backend = "hip"
        if backend == "hip" and self.hip is None:
            raise RuntimeError("the HIP encoder needs a GPU and 3- or 6-channel frames of 8..1024 pixels a side "
                               "(got device {}, shape {}x{})".format(self.device, img_shape, n_channels))
        self.img_shape, self.n_channels = tuple(img_shape), n_channels
        self._backend_arg = backend

    state_dtype = th.float32

    def replicate(self, device):
        """The same encoder on another GPU (one replica per shard of a multi-GPU HipVecEnv: the reference's single
        MultiprocessSRLModel server, rl_baselines/utils.py:162-191, becomes one per device)."""
        twin = SRLNeuralNetwork(self.state_dim, cuda=True, n_channels=self.n_channels, img_shape=self.img_shape,
                                state_dict=self.model.state_dict(), device=device, backend=self._backend_arg)
        for name in ("losses", "n_actions", "split_dimensions", "inverse_model_type"):
            if hasattr(self, name):
                setattr(twin, name, getattr(self, name))
        return twin

    def folded_weights(self):
        """((conv1_w, conv1_b), (conv2_w, conv2_b), (conv3_w, conv3_b), (fc_w, fc_b)) as float32 numpy arrays in torch
        layout with the BatchNorms folded — what srlhip_encoder_create takes."""
        convs = [m for m in self.fused_conv if isinstance(m, nn.Conv2d)]
        out = [(c.weight.detach().to("cpu", th.float32).contiguous().numpy(),
                c.bias.detach().to("cpu", th.float32).contiguous().numpy()) for c in convs]
        out.append((self.model.fc.weight.detach().to("cpu", th.float32).contiguous().numpy(),
                    self.model.fc.bias.detach().to("cpu", th.float32).contiguous().numpy()))
        return out

    @th.no_grad()
    def getStatesTorch(self, images_u8):
        """The PyTorch-ROCm forward (MIOpen / rocBLAS): reference for the fused kernel and path for other shapes."""
        if isinstance(images_u8, np.ndarray):
            images_u8 = th.from_numpy(images_u8)
        x = self.fused_conv(preprocess(images_u8.to(self.device)).contiguous(memory_format=self.memory_format))
        return self.model.fc(x.reshape(x.size(0), -1))

    @th.no_grad()
    def getStates(self, images_u8, stream=None, out=None):
        """uint8 [N][H][W][C] (numpy, or a torch tensor already on the device) -> float32 [N][state_dim].
        stream: raw HIP stream to enqueue the fused kernel on (default: torch's current stream); out: optional
        preallocated float32 [N][state_dim] device tensor (both only used by the fused HIP path)."""
        if self.hip is None:
            return self.getStatesTorch(images_u8)
        if isinstance(images_u8, np.ndarray):
            images_u8 = th.from_numpy(images_u8)
        images_u8 = images_u8.to(self.device).contiguous()
        assert images_u8.dtype == th.uint8 and tuple(images_u8.shape[1:]) == self.img_shape + (self.n_channels,), images_u8.shape
        n = images_u8.shape[0]
        if out is None:
            out = th.empty((n, self.state_dim), dtype=th.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == th.float32 and tuple(out.shape) == (n, self.state_dim)
        if stream is None:
            stream = th.cuda.current_stream(self.device).cuda_stream
        self.hip.forward(images_u8.data_ptr(), n, out.data_ptr(), stream)
        return out

    def getState(self, observation, env_id=0):
        return self.getStates(np.asarray(observation)[None])[0].to("cpu").numpy()


def getSRLDim(path=None, env_object=None):
    if path is not None:
        import json
        with open(path.rsplit("/", 1)[0] + "/exp_config.json") as f:
            return json.load(f).get("state-dim", 2)
    return env_object.getGroundTruthDim()


class SRLPCA(object):
    """state_representation/models.py:196-217 — a pickled sklearn PCA as the state representation.  The reference transforms one
    flattened uint8 observation at a time on the host (`self.model.transform(observation.reshape(-1, n_features))`); here the whole
    device-resident batch goes through ONE GEMM on the device: (X - mean_) @ components_.T [/ sqrt(explained_variance_) when the
    PCA whitens], in float64 like sklearn (uint8 input is promoted to float64 there)."""

    def __init__(self, state_dim, cuda=False, device=None):
        self.state_dim = state_dim
        self.device = th.device(device if device is not None else ("cuda" if cuda else "cpu"))
        self.model, self._w, self._b = None, None, None
        self.hip, self.backend = None, "gemm"

    state_dtype = th.float64

    def replicate(self, device):
        twin = SRLPCA(self.state_dim, device=device)
        twin.set_model(self.model)
        return twin

    def load(self, path):
        import pickle as pkl
        try:
            with open(path, "rb") as f:
                self.model = pkl.load(f)
        except UnicodeDecodeError:                 # pickle files saved with python 2 (models.py:204-207)
            with open(path, "rb") as f:
                self.model = pkl.load(f, encoding="latin1")
        self.set_model(self.model)

    def set_model(self, pca):
        self.model = pca
        comp = np.asarray(pca.components_, dtype=np.float64)                     # [state_dim][n_features]
        mean = np.zeros(comp.shape[1]) if getattr(pca, "mean_", None) is None else np.asarray(pca.mean_, dtype=np.float64)
        w = comp.T.copy()
        if getattr(pca, "whiten", False):
            w = w / np.sqrt(np.asarray(pca.explained_variance_, dtype=np.float64))[None, :]
        self._w = th.from_numpy(w).to(self.device)                               # [n_features][state_dim]
        self._b = th.from_numpy(-(mean @ w)).to(self.device)                     # folded: X @ w - mean @ w
        assert self._w.shape[1] == self.state_dim, "exp_config.json state-dim does not match the pickled PCA"

    @th.no_grad()
    def getStates(self, images_u8, stream=None, out=None):
        """uint8 [N][H][W][C] (numpy or device tensor) -> float64 [N][state_dim], one GEMM (rocBLAS dgemm on the GPU)."""
        if isinstance(images_u8, np.ndarray):
            images_u8 = th.from_numpy(images_u8)
        x = images_u8.to(self.device).reshape(images_u8.shape[0], -1).to(th.float64)
        return th.addmm(self._b, x, self._w)

    def getState(self, observation, env_id=0):
        return self.getStates(np.asarray(observation)[None])[0].to("cpu").numpy()


_SRL_HEADS = ("forward_net.", "inverse_net.", "reward_net.")


def _encoder_state_dict(state_dict, losses, split_dimensions):
    """The encoder's tensors out of a checkpoint the reference would load with `self.model.load_state_dict(th.load(path))`
    (state_representation/models.py:150-169).  There `self.model` is a bare CustomCNN only for the supervised baseline; every other
    loss combination is wrapped in srl_zoo's SRLModules (keys `model.<encoder key>` next to the forward / inverse / reward heads
    `forward_net.* / inverse_net.* / reward_net.*`), or SRLModulesSplit when `split-dimensions` is a non-zero dict.  [srl_zoo is an
    empty submodule of the reference checkout: this layout is recalled, UNVERIFIED.]  With `autoencoder` / `dae` / `vae` among the
    losses SRLModules builds CNNAutoEncoder / CNNVAE instead of CustomCNN even when model-type is `custom_cnn` — different networks
    that are not restated here: refuse instead of mis-loading."""
    other = [name for name in ("autoencoder", "dae", "vae") if name in (losses or [])]
    if other:
        raise NotImplementedError("srl model with the {} loss: srl_zoo builds CNNAutoEncoder / CNNVAE for it, only the CustomCNN encoder "
                                  "is restated here".format(" / ".join(other)))
    # (the reference takes the SRLModulesSplit path only for an OrderedDict, state_representation/models.py:159; any other value — srl_zoo
    #  is recalled to write -1 when not splitting — loads plain SRLModules)
    if isinstance(split_dimensions, dict) and sum(split_dimensions.values()) > 0:
        raise NotImplementedError("split-dimensions models (srl_zoo SRLModulesSplit) are not restated here")
    if any(k.startswith("model.") for k in state_dict):
        stray = [k for k in state_dict if not k.startswith("model.") and not k.startswith(_SRL_HEADS)]
        if stray:
            raise KeyError("unexpected keys next to the encoder in the srl checkpoint: {}".format(stray[:4]))
        return {k[len("model."):]: v for k, v in state_dict.items() if k.startswith("model.")}
    return dict(state_dict)


def loadSRLModel(path=None, cuda=False, state_dim=None, env_object=None, img_shape=(224, 224), n_channels=3, device=None):
    """state_representation/models.py:38-107, same signature and the same checks.  With a path the log folder's exp_config.json is read
    (as an OrderedDict: the order of the losses matters to srl_zoo) — `state-dim` (required), `losses`, `n_actions`, `model-type`,
    `multi-view` (-> 6 input channels: srl_zoo sets preprocessing.N_CHANNELS = 6), `inverse-model-type`, `split-dimensions` (a dict whose
    values sum to 0 means "combine the losses": None) — and the checkpoint is loaded: a `baselines/.../pca` path is a pickled PCA
    (SRLPCA), anything else a torch state_dict — a bare CustomCNN's, or srl_zoo's SRLModules layout (`model.*` + loss heads, which are
    dropped: _encoder_state_dict).  Only the `custom_cnn` encoder is restated (srl_zoo is an empty
    submodule in the reference checkout): other model types raise NotImplementedError instead of silently building a different
    network.  img_shape / n_channels: the frame shape the batched encoder is built for (the reference fixes it through srl_zoo
    globals); device: the GPU a multi-GPU caller wants this replica on (default cuda:0 when cuda)."""
    import json
    from collections import OrderedDict
    model_type, losses, n_actions, model = None, None, None, None
    use_multi_view, split_dimensions, inverse_model_type = False, None, "linear"
    if path is not None:
        log_folder = "/".join(path.split("/")[:-1]) + "/"
        with open(log_folder + "exp_config.json", "r") as f:
            exp_config = json.load(f, object_pairs_hook=OrderedDict)
        state_dim = exp_config.get("state-dim", None)
        losses = exp_config.get("losses", None)              # None for the baseline models (pca, supervised)
        n_actions = exp_config.get("n_actions", None)
        model_type = exp_config.get("model-type", None)
        use_multi_view = exp_config.get("multi-view", False)
        inverse_model_type = exp_config.get("inverse-model-type", "linear")
        assert state_dim is not None, "Please make sure you are loading an up to date model with a conform exp_config file."
        split_dimensions = exp_config.get("split-dimensions")
        if not isinstance(split_dimensions, dict) or sum(split_dimensions.values()) == 0:
            split_dimensions = None                          # combine the losses instead of splitting (a dict summing to 0, or no dict: -1 / None)
    else:
        assert env_object is not None or (state_dim is not None and state_dim > 0), \
            "When learning states, state_dim must be > 0. Otherwise, set SRL_MODEL_PATH to a srl_model.pth file with learned states."
        model_type = "custom_cnn"                            # (the reference builds the model that is being learned: srl_zoo's default)
        losses, n_actions = [], 0
    if path is not None and "baselines" in path and "pca" in path:
        model_type = "pca"
        model = SRLPCA(state_dim, cuda, device=device)
    assert model_type is not None or model is not None, \
        "Model type not supported. In order to use loadSRLModel, a path to an SRL model must be given."
    assert not (losses is None and not model_type == "pca"), \
        "Please make sure you are loading an up to date model with a conform exp_config file."
    assert not (n_actions is None and not (model_type == "pca" or "supervised" in losses)), \
        "Please make sure you are loading an up to date model with a conform exp_config file."
    if model is None:
        if use_multi_view:
            n_channels = 6                                   # preprocessing.preprocess.N_CHANNELS = 6
        if model_type != "custom_cnn":
            raise NotImplementedError("srl model type {!r}: only srl_zoo's custom_cnn encoder (and the pca baseline) are restated here "
                                      "(srl_zoo is an empty submodule of the reference checkout)".format(model_type))
        state_dict = th.load(path, map_location="cpu") if path is not None else None
        if isinstance(state_dict, dict) and "state_dict" in state_dict:
            state_dict = state_dict["state_dict"]
        if state_dict is not None:
            state_dict = _encoder_state_dict(state_dict, losses, split_dimensions)
        model = SRLNeuralNetwork(state_dim, cuda, model_type, n_channels=n_channels, img_shape=img_shape, state_dict=state_dict, device=device)
        model.losses, model.n_actions, model.split_dimensions, model.inverse_model_type = losses, n_actions, split_dimensions, inverse_model_type
    elif path is not None:
        model.load(path)
    return model
