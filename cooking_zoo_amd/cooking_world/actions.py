"""Integer action codes (reference: cooking_world/actions.py:2-50).

scheme3 = 5 actions, "walk into a thing to use it"; scheme1 = 8 actions with explicit interact codes.
scheme2 is dead in the reference (action_scheme2.py:15 calls a method CookingWorld does not define)
and is therefore not offered.
"""


class ActionScheme1:
    NO_OP, WALK_LEFT, WALK_RIGHT, WALK_DOWN, WALK_UP = 0, 1, 2, 3, 4
    INTERACT_PRIMARY, INTERACT_PICK_UP_SPECIAL, EXECUTE_ACTION = 5, 6, 7
    WALK_ACTIONS = [WALK_UP, WALK_DOWN, WALK_RIGHT, WALK_LEFT]
    INTERACT_ACTIONS = [INTERACT_PRIMARY, INTERACT_PICK_UP_SPECIAL, EXECUTE_ACTION]
    ACTIONS = [NO_OP, WALK_LEFT, WALK_RIGHT, WALK_DOWN, WALK_UP, INTERACT_PRIMARY, INTERACT_PICK_UP_SPECIAL,
               EXECUTE_ACTION]
    CODE = 1


class ActionScheme3:
    NO_OP, WALK_LEFT, WALK_RIGHT, WALK_DOWN, WALK_UP = 0, 1, 2, 3, 4
    WALK_ACTIONS = [WALK_UP, WALK_DOWN, WALK_RIGHT, WALK_LEFT]
    INTERACT_ACTIONS = []
    ACTIONS = [NO_OP, WALK_LEFT, WALK_RIGHT, WALK_DOWN, WALK_UP]
    CODE = 3


ACTION_SCHEMES = {"scheme1": ActionScheme1, "scheme3": ActionScheme3}
