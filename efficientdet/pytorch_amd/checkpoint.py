"""Checkpoint format and offline pretrained-backbone loader.

* ``get_state_dict`` -- utils/helper.py:25-30, extended to DistributedDataParallel (the reference only unwraps
  nn.DataParallel, so its DDP checkpoints carry a ``module.`` prefix that its own ``load_state_dict`` then rejects).
* ``save_checkpoint`` / ``load_checkpoint`` -- the dict of train.py:279-291 ({'epoch', 'parser', 'state_dict'}) plus the
  optimizer state the reference forgets, ``module.``-prefix tolerant on load (train.py:212-236).
* ``load_pretrained_backbone`` -- models/utils.py:305-328 (``load_pretrained_weights``) without the network: the
  ``efficientnet-b*.pth`` file named by ``url_map`` is looked up in a local directory; same key handling (drop ``_fc.*``
  unless load_fc, and insist that nothing else is missing)."""
import os

import torch


# file names of models/utils.py:305-314 (the basename of each URL)
PRETRAINED_FILES = {
    'efficientnet-b0': 'efficientnet-b0-355c32eb.pth', 'efficientnet-b1': 'efficientnet-b1-f1951068.pth',
    'efficientnet-b2': 'efficientnet-b2-8bb594d6.pth', 'efficientnet-b3': 'efficientnet-b3-5fb5a3c3.pth',
    'efficientnet-b4': 'efficientnet-b4-6ed6700e.pth', 'efficientnet-b5': 'efficientnet-b5-b6417697.pth',
    'efficientnet-b6': 'efficientnet-b6-c76e70fd.pth', 'efficientnet-b7': 'efficientnet-b7-dcc49843.pth',
}


def unwrap(model):
    while isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        model = model.module
    return model


def get_state_dict(model):
    return unwrap(model).state_dict()


def strip_module_prefix(sd):
    return {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}


def save_checkpoint(path, model, epoch, args=None, optimizer=None):
    state = {'epoch': epoch, 'parser': args, 'state_dict': get_state_dict(model)}
    if optimizer is not None:
        state['optimizer'] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, path)
    return state


def load_checkpoint(path, model=None, optimizer=None, map_location='cpu'):
    """-> the checkpoint dict; loads model / optimizer state when given (strict, prefix-tolerant)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    if model is not None:
        unwrap(model).load_state_dict(strip_module_prefix(ck['state_dict']))
    if optimizer is not None and 'optimizer' in ck:
        optimizer.load_state_dict(ck['optimizer'])
    return ck


def load_pretrained_backbone(model, source=None, load_fc=False, directory=None, reference_semantics=False):
    """Load ImageNet EfficientNet weights into ``model.backbone`` (what ``EfficientNet.from_pretrained`` does in the
    reference, models/efficientnet.py:222-226 -> models/utils.py:317-328).  ``source``: a state_dict, a .pth path, or None =
    ``PRETRAINED_FILES[backbone name]`` inside ``directory`` / $EFFDET_PRETRAINED_DIR / ~/.cache/torch/hub/checkpoints.

    ``reference_semantics``: the reference calls ``from_pretrained`` FIRST and then runs its initialisation loop over
    ``self.modules()`` (models/efficientdet.py:33, 47-53), which redraws EVERY ``nn.Conv2d`` weight -- the backbone's included --
    from N(0, sqrt(2/n)) and resets every BatchNorm weight / bias to 1 / 0.  What survives of the pretrained file in the
    reference's model is therefore only: the BatchNorm running statistics, the conv BIASES (the squeeze-excite convs) and the
    unused ``_fc``.  ``False`` (default) keeps the loaded weights -- what a user loading pretrained weights wants;
    ``True`` reproduces the reference's effective behaviour by re-running that loop over the backbone after loading."""
    m = unwrap(model)
    name = m.backbone.model_name
    if source is None:
        dirs = [directory, os.environ.get('EFFDET_PRETRAINED_DIR'), os.path.expanduser('~/.cache/torch/hub/checkpoints')]
        cands = [os.path.join(d, PRETRAINED_FILES[name]) for d in dirs if d and name]
        source = next((c for c in cands if os.path.isfile(c)), None)
        if source is None:
            raise FileNotFoundError('no offline copy of %s (looked in %s); this build has no network access by design'
                                    % (PRETRAINED_FILES.get(name), [d for d in dirs if d]))
    sd = torch.load(source, map_location='cpu', weights_only=True) if isinstance(source, (str, os.PathLike)) else dict(source)
    sd = strip_module_prefix(sd)
    if load_fc:
        m.backbone.load_state_dict(sd)
    else:
        sd.pop('_fc.weight', None); sd.pop('_fc.bias', None)
        res = m.backbone.load_state_dict(sd, strict=False)
        assert set(res.missing_keys) == {'_fc.weight', '_fc.bias'}, 'issue loading pretrained weights'
        assert not res.unexpected_keys, res.unexpected_keys
    if reference_semantics:
        import math
        for mod in m.backbone.modules():                        # models/efficientdet.py:47-53, restricted to the backbone
            if isinstance(mod, torch.nn.Conv2d):
                n = mod.kernel_size[0] * mod.kernel_size[1] * mod.out_channels
                mod.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data.fill_(1)
                mod.bias.data.zero_()
    m._prep = {}
    return name
