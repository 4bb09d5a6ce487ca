"""State enums of the food / tool state machines the hot path uses.

Mirrors the three enums that are live in the reference (cooking_world/constants.py:4-12,45-47);
the reference's Toaster/Microwave/Pot/Temperature enums have no concrete object class and are out of scope.
On the device these collapse to single bits (soa.DYN_CHOPPED, soa.DYN_MASHED, soa.CELL_READY).
"""
from enum import Enum


class ChopFoodStates(Enum):
    FRESH = "Fresh"
    CHOPPED = "Chopped"


class BlenderFoodStates(Enum):
    FRESH = "Fresh"
    IN_PROGRESS = "InProgress"   # transient inside BlenderFood.blend only; never observable between steps
    MASHED = "Mashed"


class ActionObjectState(Enum):
    READY = "Ready"
    NOT_USABLE = "NotUsable"
