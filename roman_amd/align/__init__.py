"""roman_amd.align — the reference's registration plugin surface (roman.align) on MI355X.

Same class names, constructor arguments, return values and exceptions as the reference
([REF roman/align/object_registration.py], [REF roman/align/roman_registration.py],
[REF roman/align/dist_reg_with_pruning.py], [REF roman/params/submap_align_params.py:86-150]), plus
the batched entry points the reference lacks (its pair loop is serial:
[REF roman/align/submap_align.py:93-200]).
"""
from .object_registration import InsufficientAssociationsException, ObjectRegistration  # noqa: F401
from .roman_registration import FusionMethod, ROMANParams, ROMANRegistration  # noqa: F401
from .dist_reg_with_pruning import DistRegWithPruning, GravityConstraintError  # noqa: F401
from .submap_align_params import SubmapAlignParams  # noqa: F401
from .batch import AlignmentBatch, align_pairs, all_pairs_problems  # noqa: F401
from .submap_align import Submap, SubmapAlignIO, SubmapAlignResults  # noqa: F401  (the loop: roman_amd.align.submap_align.submap_align)
