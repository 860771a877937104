"""Hooks between the hand-scheduled backward regions (ViT encoder, VLG head, ResNetV1c side encoder) and the data-parallel
gradient reducer (train.GradAllReducer): the replacement of torch DDP's autograd-hook reducer (semivl.py:139-140, fired
from inside `loss.backward()`, semivl.py:327).

A region's forward calls `expect(params)` when it records itself for backward; its backward calls `ready(params)` as soon
as it has accumulated its contribution to those parameters' `main_grad` arena views.  When every expected contribution of
every parameter of a bucket has arrived, the reducer launches that bucket's all-reduce on its communication stream while
the rest of the backward keeps running on the compute stream.  Without an attached reducer (single process) both calls
are no-ops."""


def expect(params):
    for p in params:
        r = getattr(p, "_svl_reducer", None)
        if r is not None:
            r._expect(p)


def ready(params):
    for p in params:
        r = getattr(p, "_svl_reducer", None)
        if r is not None:
            r._ready(p)
