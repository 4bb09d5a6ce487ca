"""The recipe book: 26 default goal nodes and 8 named recipes, plus the user registry.

Same names, goal ids and registry behaviour as the reference (cooking_book/recipe_drawer.py:19-35,
40-118; table in SURVEY.md B.2).  `RECIPES[name]()` returns a fresh `Recipe`.
"""
from copy import deepcopy

from cooking_zoo_amd.cooking_book.recipe import Recipe, RecipeNode
from cooking_zoo_amd.cooking_world.constants import ChopFoodStates, BlenderFoodStates

NUM_GOALS = 0
DEFAULT_NUM_GOALS = 0
RECIPE_STORE = {}
_user_ids = iter(range(1 << 30))
_default_ids = iter(range(1 << 30))


def get_next_id():
    global NUM_GOALS
    NUM_GOALS += 1
    return next(_user_ids)


def get_next_default_id():
    global DEFAULT_NUM_GOALS
    DEFAULT_NUM_GOALS += 1
    return next(_default_ids)


def register_recipe(recipe, name):
    RECIPE_STORE[name] = lambda: deepcopy(recipe)


def _leaf(name, attr, value):
    return RecipeNode(root_type=name, id_num=get_next_default_id(), name=name, conditions=[(attr, value)])


_CH = ("chop_state", ChopFoodStates.CHOPPED)
_MA = ("blend_state", BlenderFoodStates.MASHED)
# goal ids 0..9: prepared ingredients
ChoppedLettuce = _leaf("Lettuce", *_CH)
ChoppedOnion = _leaf("Onion", *_CH)
ChoppedTomato = _leaf("Tomato", *_CH)
ChoppedApple = _leaf("Apple", *_CH)
ChoppedCucumber = _leaf("Cucumber", *_CH)
ChoppedWatermelon = _leaf("Watermelon", *_CH)
ChoppedBanana = _leaf("Banana", *_CH)
MashedBanana = _leaf("Banana", *_MA)
ChoppedCarrot = _leaf("Carrot", *_CH)
MashedCarrot = _leaf("Carrot", *_MA)


def _plate(*contains):
    return RecipeNode(root_type="Plate", id_num=get_next_default_id(), name="Plate", contains=list(contains))


# goal ids 10..16: plated combinations
TomatoSaladPlate = _plate(ChoppedTomato)
TomatoLettucePlate = _plate(ChoppedTomato, ChoppedLettuce)
TomatoLettuceOnionPlate = _plate(ChoppedTomato, ChoppedLettuce, ChoppedOnion)
CarrotBananaPlate = _plate(ChoppedCarrot, ChoppedBanana)
MashedCarrotBananaPlate = _plate(MashedCarrot, MashedBanana)
CucumberOnionPlate = _plate(ChoppedCucumber, ChoppedOnion)
AppleWatermelonPlate = _plate(ChoppedApple, ChoppedWatermelon)


def _delivered(plate):
    return RecipeNode(root_type="Deliversquare", id_num=get_next_default_id(), name="Deliversquare", contains=[plate])


# goal ids 17..23: delivered dishes
TomatoSalad = _delivered(TomatoSaladPlate)
TomatoLettuceSalad = _delivered(TomatoLettucePlate)
TomatoLettuceOnionSalad = _delivered(TomatoLettuceOnionPlate)
CarrotBanana = _delivered(CarrotBananaPlate)
MashedCarrotBanana = _delivered(MashedCarrotBananaPlate)
CucumberOnion = _delivered(CucumberOnionPlate)
AppleWatermelon = _delivered(AppleWatermelonPlate)
# goal ids 24, 25: the unreachable placeholder task
floor = RecipeNode(root_type="Floor", id_num=get_next_default_id(), name="Floor")
no_recipe_node = RecipeNode(root_type="Deliversquare", id_num=get_next_default_id(), name="Deliversquare",
                            contains=[floor])


def _maker(root, name):
    return lambda: deepcopy(Recipe(root, DEFAULT_NUM_GOALS, name))


RECIPES = {name: _maker(root, name) for name, root in [
    ("TomatoSalad", TomatoSalad), ("TomatoLettuceSalad", TomatoLettuceSalad), ("CarrotBanana", CarrotBanana),
    ("MashedCarrotBanana", MashedCarrotBanana), ("CucumberOnion", CucumberOnion),
    ("AppleWatermelon", AppleWatermelon), ("TomatoLettuceOnionSalad", TomatoLettuceOnionSalad),
    ("no_recipe", no_recipe_node)]}
