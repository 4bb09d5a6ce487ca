/*
 * cz_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar, single-threaded CPU restatement of the reference CookingZoo step() hot path
 * (DavidRother/cooking_zoo @ 2024-10-16).  It exists only to check the HIP kernels:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (cooking_zoo_amd + libcookingzoo_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py (and test_spawn_keyed.py) replay every golden trace under
 * tests/golden/ (captured by tools/gen_golden.py from the unmodified reference imported in the
 * build container, incl. SURVEY.md Appendix C.3's known-answer trace) and requires bit-equality
 * of state, float64 observations, float64 rewards and flags at every step.
 *
 * Style: deliberately object/list based like the reference (explicit `content` lists, linear
 * scans in class-key order) and NOT like the GPU kernel (ballots over lanes), so that the two are
 * independent derivations of the same semantics.  Every function cites the reference lines it follows
 * (paths relative to /root/reference/cooking_zoo/).
 *
 * Record format: cooking_zoo_amd/soa.py (include/cookingzoo.h restates it for C callers).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CZO_MAX_AGENTS 4
#define CZO_MAX_NODES 16          /* recipe.py puts no limit on a graph; the flat tables carry up to 16 nodes */
#define CZO_MAX_RECIPES 4
#define CZO_MAX_CELLS 1024
#define CZO_MAX_DYN 255
#define CZO_PLATE_MAX 64

enum { FLOOR, COUNTER, DELIVERSQUARE, SWITCH, BLOCK, CUTBOARD, BLENDER };
enum { PLATE, ONION, TOMATO, LETTUCE, CARROT, CUCUMBER, BANANA, APPLE, WATERMELON, BREAD };
enum { CELL_READY = 8, CELL_TOGGLE = 16, CELL_ACTIVE = 32, CELL_WALK = 64 };
enum { DYN_ALIVE = 1, DYN_CHOPPED = 2, DYN_MASHED = 4, DYN_FREE = 8 };
enum { COND_NONE, COND_CHOPPED, COND_MASHED, COND_NOT_CHOPPED, COND_NOT_MASHED };
enum { W_T, W_MARKS, W_LAYOUT, W_STATUS, W_EPISODE, W_RECIPES, W_POOL, W_RES1, HDR_WORDS };
enum { STATUS_DONE = 1, STATUS_TERM = 2, STATUS_TRUNC = 4 };

typedef struct {
    int32_t width, height, max_dyn, num_agents, feat_len;
    int32_t action_scheme;        /* 1 or 3 */
    int32_t max_steps;
    int32_t end_condition_all;
    int32_t num_recipes;          /* recipe graphs per env (>= num_agents, <= 4) */
    int32_t auto_reset;           /* 0: done envs freeze; 1: next-step reset from the layout pool */
    int32_t num_layouts;
    int32_t record_words;
    int32_t recipe_nodes;         /* node capacity of a recipe-table row: 8 (rows of 9 words, marks 8 bits per recipe in
                                     word 1) or 16 (rows of 33 words, marks 16 bits per recipe in words 1 and 7) */
    double recipe_reward, max_time_penalty, recipe_penalty, recipe_node_reward;
} czo_config;

/* one layout of the pool: what reset() needs and what the observation order depends on */
typedef struct {
    const uint32_t *init_record;   /* record_words words: initial state (t=0, marks ignored)   */
    const int32_t *static_off;     /* [8] offsets into static_cells per static type (list order) */
    const int16_t *static_cells;   /* cell indices                                              */
} czo_layout;

typedef struct { int cls, num; } czo_meta_entry;   /* cls: 0..6 static, 16..25 dynamic, 32 = Agent */

/* ------------------------------------------------------------------------------------------ */
/* object model (unpacked record)                                                              */

typedef struct {
    int alive, cls, x, y, chopped, mashed, free_;
    int container;                 /* plate slot or -1 */
    int content[CZO_PLATE_MAX];    /* Plate.content  (world_objects.py:386) */
    int ncontent;
} Dyn;

typedef struct {
    int type, ready, toggle, active, walk, pressed;
    int content[8];                /* dynamic slots directly on this static (agents are not tracked) */
    int ncontent;
} Stat;

typedef struct { int x, y, orient, holding; } Agent;

typedef struct {
    const czo_config *cfg;
    int W, H, D, A;
    Stat cell[CZO_MAX_CELLS];
    Dyn obj[CZO_MAX_DYN];
    Agent ag[CZO_MAX_AGENTS];
    int err;
} World;

static int cell_off(const czo_config *c) { return HDR_WORDS + CZO_MAX_AGENTS + 2 * CZO_MAX_AGENTS; /* + running returns */ }
static int dyn0_off(const czo_config *c) { return cell_off(c) + (c->width * c->height + 3) / 4; }
static int dyn1_off(const czo_config *c) { return dyn0_off(c) + c->max_dyn; }

static void unpack(World *w, const czo_config *cfg, const uint32_t *rec)
{
    w->cfg = cfg; w->W = cfg->width; w->H = cfg->height; w->D = cfg->max_dyn; w->A = cfg->num_agents; w->err = 0;
    const uint8_t *cb = (const uint8_t *)(rec + cell_off(cfg));
    for (int c = 0; c < w->W * w->H; ++c) {
        Stat *s = &w->cell[c];
        s->type = cb[c] & 7; s->ready = !!(cb[c] & CELL_READY); s->toggle = !!(cb[c] & CELL_TOGGLE);
        s->active = !!(cb[c] & CELL_ACTIVE); s->walk = !!(cb[c] & CELL_WALK); s->pressed = 0; s->ncontent = 0;
    }
    for (int a = 0; a < CZO_MAX_AGENTS; ++a) {
        uint32_t v = rec[HDR_WORDS + a];
        w->ag[a].x = v & 255; w->ag[a].y = (v >> 8) & 255; w->ag[a].orient = (v >> 16) & 255;
        w->ag[a].holding = (int)((v >> 24) & 255) - 1;
    }
    for (int s = 0; s < w->D; ++s) {
        uint32_t a = rec[dyn0_off(cfg) + s], b = rec[dyn1_off(cfg) + s];
        Dyn *o = &w->obj[s];
        int fl = a >> 24;
        o->x = a & 255; o->y = (a >> 8) & 255; o->cls = (a >> 16) & 255;
        o->alive = !!(fl & DYN_ALIVE); o->chopped = !!(fl & DYN_CHOPPED); o->mashed = !!(fl & DYN_MASHED);
        o->free_ = !!(fl & DYN_FREE);
        o->container = (int)(b & 255) - 1; o->ncontent = 0;
    }
    /* plate content lists in stored sequence order */
    for (int s = 0; s < w->D; ++s) {
        Dyn *o = &w->obj[s];
        if (!o->alive || o->container < 0) continue;
        int seq = (rec[dyn1_off(cfg) + s] >> 8) & 255;
        Dyn *p = &w->obj[o->container];
        if (seq >= CZO_PLATE_MAX) { w->err = 1; continue; }
        p->content[seq] = s;
        if (seq + 1 > p->ncontent) p->ncontent = seq + 1;
    }
    /* static content = objects on the cell that are neither in a plate nor held, in slot order
       (soa.py header; asserted against the reference by tools/gen_golden.py) */
    for (int s = 0; s < w->D; ++s) {
        Dyn *o = &w->obj[s];
        if (!o->alive || o->container >= 0) continue;
        int held = 0;
        for (int a = 0; a < w->A; ++a) if (w->ag[a].holding == s) held = 1;
        if (held) continue;
        Stat *st = &w->cell[o->y * w->W + o->x];
        if (st->ncontent < 8) st->content[st->ncontent++] = s; else w->err = 2;
    }
}

static void pack(World *w, uint32_t *rec)
{
    const czo_config *cfg = w->cfg;
    uint8_t *cb = (uint8_t *)(rec + cell_off(cfg));
    for (int c = 0; c < w->W * w->H; ++c) {
        Stat *s = &w->cell[c];
        cb[c] = (uint8_t)(s->type | (s->ready ? CELL_READY : 0) | (s->toggle ? CELL_TOGGLE : 0) |
                          (s->active ? CELL_ACTIVE : 0) | (s->walk ? CELL_WALK : 0));
    }
    for (int a = 0; a < w->A; ++a)
        rec[HDR_WORDS + a] = (uint32_t)w->ag[a].x | ((uint32_t)w->ag[a].y << 8) | ((uint32_t)w->ag[a].orient << 16) |
                             ((uint32_t)((w->ag[a].holding + 1) & 255) << 24);
    for (int s = 0; s < w->D; ++s) {
        Dyn *o = &w->obj[s];
        int fl = (o->alive ? DYN_ALIVE : 0) | (o->chopped ? DYN_CHOPPED : 0) | (o->mashed ? DYN_MASHED : 0) |
                 (o->free_ ? DYN_FREE : 0);
        rec[dyn0_off(cfg) + s] = (uint32_t)o->x | ((uint32_t)o->y << 8) | ((uint32_t)o->cls << 16) | ((uint32_t)fl << 24);
        rec[dyn1_off(cfg) + s] = 0;
    }
    for (int p = 0; p < w->D; ++p) {
        Dyn *pl = &w->obj[p];
        if (!pl->alive) continue;
        for (int i = 0; i < pl->ncontent; ++i) {
            int s = pl->content[i];
            if (w->obj[s].container != p) w->err = 3;
            rec[dyn1_off(cfg) + s] = (uint32_t)(p + 1) | ((uint32_t)i << 8);
        }
    }
    /* the flat model can only express static content that is in slot order: verify */
    for (int c = 0; c < w->W * w->H; ++c) {
        Stat *st = &w->cell[c];
        for (int i = 0; i < st->ncontent; ++i) {
            Dyn *o = &w->obj[st->content[i]];
            if (o->container >= 0 || o->y * w->W + o->x != c) w->err = 4;
            if (i && st->content[i - 1] > st->content[i]) w->err = 5;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* helpers mirroring the reference's queries                                                   */

static int is_action_object(int type) { return type == CUTBOARD || type == BLENDER; }       /* world_objects.py:242,314 */
static int is_blender_food(int cls) { return cls == CARROT || cls == BANANA; }              /* :543,:616 */
static int is_food(int cls) { return cls != PLATE; }
static int done_(const Dyn *o) { return o->chopped || o->mashed; }                          /* :441,:549,:622 */

/* cooking_world.py:223-227 square_walkable */
static int walkable(const World *w, int x, int y)
{
    const Stat *s = &w->cell[y * w->W + x];
    if (s->type == FLOOR || s->type == SWITCH) return 1;
    if (s->type == BLOCK) return s->walk;
    return 0;
}

/* cooking_world.py:232-241 get_objects_at(location, DynamicObject): class-key order then list order == slot order */
static int dyn_at(const World *w, int x, int y, int *out)
{
    int n = 0;
    for (int s = 0; s < w->D; ++s)
        if (w->obj[s].alive && w->obj[s].x == x && w->obj[s].y == y) out[n++] = s;
    return n;
}

/* cooking_world.py:172-184 get_target_location */
static void target(int x, int y, int action, int *tx, int *ty)
{
    *tx = x; *ty = y;
    if (action == 1) *tx = x - 1; else if (action == 2) *tx = x + 1;
    else if (action == 3) *ty = y + 1; else if (action == 4) *ty = y - 1;
}

/* Plate.move_to world_objects.py:393-396 / Object.move_to abstract_classes.py:20 */
static void obj_move_to(World *w, int s, int x, int y)
{
    Dyn *o = &w->obj[s];
    if (o->cls == PLATE)
        for (int i = 0; i < o->ncontent; ++i) { w->obj[o->content[i]].x = x; w->obj[o->content[i]].y = y; }
    o->x = x; o->y = y;
}

static void content_remove(Stat *st, int s)
{
    int k = -1;
    for (int i = 0; i < st->ncontent; ++i) if (st->content[i] == s) { k = i; break; }
    if (k < 0) return;
    for (int i = k; i + 1 < st->ncontent; ++i) st->content[i] = st->content[i + 1];
    st->ncontent--;
}

static int content_has(const Stat *st, int s)
{
    for (int i = 0; i < st->ncontent; ++i) if (st->content[i] == s) return 1;
    return 0;
}

/* "for c in content: c.free = False; content[-1].free = True"  (world_objects.py:73-75 and twins) */
static void static_append(World *w, Stat *st, int s)
{
    if (st->ncontent < 8) st->content[st->ncontent++] = s; else w->err = 6;
    for (int i = 0; i < st->ncontent; ++i) w->obj[st->content[i]].free_ = 0;
    w->obj[st->content[st->ncontent - 1]].free_ = 1;
}

/* Plate.accepts world_objects.py:408-409 */
static int plate_accepts(const World *w, int p, int s)
{
    const Dyn *o = &w->obj[s];
    return is_food(o->cls) && done_(o) && w->obj[p].ncontent < CZO_PLATE_MAX;
}

/* Plate.add_content world_objects.py:398-406 */
static void plate_add(World *w, int p, int s)
{
    Dyn *pl = &w->obj[p];
    pl->content[pl->ncontent++] = s;
    w->obj[s].container = p;
    for (int i = 0; i < pl->ncontent; ++i) w->obj[pl->content[i]].free_ = 0;
    w->obj[pl->content[pl->ncontent - 1]].free_ = 1;
}

/* StaticObject.accepts per class: world_objects.py:23,64-66,107-108,153,208,271-273,337-338 */
static int static_accepts(const World *w, const Stat *st, int s)
{
    const Dyn *o = &w->obj[s];
    switch (st->type) {
    case COUNTER: return st->ncontent < 1;
    case DELIVERSQUARE: return st->ncontent < 1;
    case CUTBOARD: return is_food(o->cls) /* every food class is a ChopFood */ && st->ncontent < 1 && !o->chopped;
    case BLENDER: return is_blender_food(o->cls) && !st->toggle && st->ncontent + 1 <= 1 && !o->mashed;
    default: return 0;
    }
}

/* add_content per class: world_objects.py:71-75,110-115,280-288,348-354 */
static void static_add(World *w, Stat *st, int s)
{
    switch (st->type) {
    case COUNTER: static_append(w, st, s); break;
    case DELIVERSQUARE: if (static_accepts(w, st, s)) static_append(w, st, s); break;
    case CUTBOARD: if (static_accepts(w, st, s)) { st->ready = 1; static_append(w, st, s); } break;
    case BLENDER: if (static_accepts(w, st, s)) { st->ready = 1; static_append(w, st, s); } break;
    default: break;
    }
}

/* releases per class: world_objects.py:26,68,117-118,156,202,275-278,340-346 */
static int static_releases(Stat *st)
{
    switch (st->type) {
    case DELIVERSQUARE: return 0;
    case CUTBOARD: if (st->ncontent == 1) st->ready = 0; return 1;
    case BLENDER: {
        int valid = !st->toggle;
        if (valid && st->ncontent - 1 == 0) st->ready = 0;
        return valid;
    }
    default: return 1;
    }
}

static int agent_at(const World *w, int x, int y)
{
    for (int a = 0; a < w->A; ++a) if (w->ag[a].x == x && w->ag[a].y == y) return 1;
    return 0;
}

static int in_bounds(const World *w, int x, int y) { return x >= 0 && y >= 0 && x < w->W && y < w->H; }

/* Agent.grab world_objects.py:785-787 */
static void agent_grab(World *w, Agent *ag, int s)
{
    ag->holding = s;
    obj_move_to(w, s, ag->x, ag->y);
}

/* Agent.put_down world_objects.py:789-791 */
static void agent_put_down(World *w, Agent *ag, int x, int y)
{
    obj_move_to(w, ag->holding, x, y);
    ag->holding = -1;
}

/* cooking_world.py:243-261 attempt_merge */
static void attempt_merge(World *w, Agent *ag, const int *dyn, int ndyn, int lx, int ly, Stat *st)
{
    int plates[CZO_MAX_DYN], np = 0;
    for (int i = 0; i < ndyn; ++i) if (w->obj[dyn[i]].cls == PLATE) plates[np++] = dyn[i];
    int held = ag->holding;
    if (np == 1) {
        if (plate_accepts(w, plates[0], held)) {
            plate_add(w, plates[0], held);
            agent_put_down(w, ag, lx, ly);
        }
    } else if (w->obj[held].cls == PLATE && ndyn > 0) {
        int pick = dyn[ndyn - 1];
        if (plate_accepts(w, held, pick)) {
            plate_add(w, held, pick);
            obj_move_to(w, pick, ag->x, ag->y);
            if (!content_has(st, pick)) w->err = 7;      /* reference would raise ValueError */
            content_remove(st, pick);
        }
    } else {
        if (static_accepts(w, st, held)) {
            static_add(w, st, held);
            agent_put_down(w, ag, lx, ly);
        }
    }
}

/* cooking_world.py:114-136 resolve_primary_interaction */
static void resolve_primary_interaction(World *w, Agent *ag)
{
    int lx, ly;
    target(ag->x, ag->y, ag->orient, &lx, &ly);
    if (!in_bounds(w, lx, ly)) return;           /* reference: IndexError (scheme1, facing off-grid); build: no-op */
    if (agent_at(w, lx, ly)) return;
    int dyn[CZO_MAX_DYN];
    int ndyn = dyn_at(w, lx, ly, dyn);
    Stat *st = &w->cell[ly * w->W + lx];
    if (ag->holding < 0 && ndyn == 0) return;
    if (ag->holding < 0) {
        if (static_releases(st)) {
            int grab = dyn[ndyn - 1];
            for (int i = 0; i < ndyn; ++i) if (w->obj[dyn[i]].free_) { grab = dyn[i]; break; }
            if (content_has(st, grab)) {
                agent_grab(w, ag, grab);
                content_remove(st, grab);
            }
        }
    } else {
        attempt_merge(w, ag, dyn, ndyn, lx, ly, st);
    }
}

/* cooking_world.py:138-154 resolve_interaction_pick_up_special (scheme1) */
static void resolve_pick_up_special(World *w, Agent *ag)
{
    int lx, ly;
    target(ag->x, ag->y, ag->orient, &lx, &ly);
    if (!in_bounds(w, lx, ly)) return;
    if (agent_at(w, lx, ly)) return;
    int dyn[CZO_MAX_DYN];
    int ndyn = dyn_at(w, lx, ly, dyn);
    if (ag->holding >= 0 || ndyn == 0) return;
    int plate = -1, np = 0;
    for (int i = 0; i < ndyn; ++i) if (w->obj[dyn[i]].cls == PLATE) { plate = dyn[i]; np++; }
    if (np != 1) return;
    Dyn *pl = &w->obj[plate];
    if (pl->ncontent == 0) return;               /* IndexError swallowed */
    int s = pl->content[--pl->ncontent];
    w->obj[s].container = -1;
    agent_grab(w, ag, s);
}

/* cooking_world.py:156-170 resolve_execute_action + Cutboard.action :250-269 + Bread.chop :738-745
   + Blender.action :356-360 */
static void resolve_execute_action(World *w, Agent *ag)
{
    int lx, ly;
    target(ag->x, ag->y, ag->orient, &lx, &ly);
    if (!in_bounds(w, lx, ly)) return;
    if (agent_at(w, lx, ly)) return;
    Stat *st = &w->cell[ly * w->W + lx];
    if (st->type == CUTBOARD) {
        if (!st->ready) return;
        /* REFERENCE CRASH, reachable: a READY board whose loop executes nothing - empty content (a mashed Banana was put on it,
           accepts() :270-272 only looks at chop_state, and a Plate in hand absorbed it through attempt_merge branch 2,
           cooking_world.py:250-256, which removes it from `content` without releases()) - falls off Cutboard.action, returns None,
           and cooking_world.py:162 raises TypeError.  Build: nothing created, deleted or executed; the board stays READY.
           Pinned by tests/golden/refcrash_cutboard_scheme1.npz (reference with that raise site as this no-op). */
        for (int i = 0; i < st->ncontent; ++i) {
            Dyn *o = &w->obj[st->content[i]];
            /* ChopFood.chop abstract_classes.py:250-254 */
            if (o->chopped) continue;            /* action_executed False -> the loop goes on and falls off its end */
            o->chopped = 1;
            if (o->cls == BREAD) {
                /* new chopped Bread at the same cell, appended to the board content and to world_objects["Bread"]:
                   first not-alive Bread slot (clone head-room follows the originals) */
                int c = -1;
                for (int s = 0; s < w->D; ++s) if (!w->obj[s].alive && w->obj[s].cls == BREAD) { c = s; break; }
                if (c < 0) { w->err = 8; }
                else {
                    Dyn *n = &w->obj[c];
                    n->alive = 1; n->chopped = 1; n->mashed = 0; n->free_ = 1; n->container = -1; n->ncontent = 0;
                    n->x = o->x; n->y = o->y;
                    if (st->ncontent < 8) st->content[st->ncontent++] = c;
                }
            }
            st->ready = 0;
            return;
        }
    } else if (st->type == BLENDER) {
        if (st->ready) st->toggle = !st->toggle;
    }
}

/* action_scheme3.py:26-34 / action_scheme1.py:23-31 resolve_walking_action; returns 1 if the agent "moved" */
static int resolve_walking_action(World *w, Agent *ag, int action)
{
    int tx, ty;
    target(ag->x, ag->y, action, &tx, &ty);
    if (!walkable(w, tx, ty)) return 0;
    ag->x = tx; ag->y = ty;                                  /* Agent.move_to world_objects.py:793-796 */
    if (ag->holding >= 0) obj_move_to(w, ag->holding, tx, ty);
    Stat *t = &w->cell[ty * w->W + tx];
    if (t->type == SWITCH) { t->active = !t->active; t->pressed = 1; }   /* Switch.add_content :159-163 */
    return 1;
}

/* action_scheme3.py:37-43 resolve_interaction */
static void scheme3_interaction(World *w, Agent *ag, int tx, int ty)
{
    Stat *st = &w->cell[ty * w->W + tx];
    int dyn[CZO_MAX_DYN];
    int ndyn = dyn_at(w, tx, ty, dyn);
    int any_not_done = 0;
    for (int i = 0; i < ndyn; ++i) if (!done_(&w->obj[dyn[i]])) any_not_done = 1;
    if (is_action_object(st->type) && any_not_done) resolve_execute_action(w, ag);
    else resolve_primary_interaction(w, ag);
}

/* cooking_world.py:192-204 check_inbounds */
static void check_inbounds(const World *w, const int *actions, int *out)
{
    for (int a = 0; a < w->A; ++a) {
        int act = actions[a];
        if (act < 0 || act == 0 || act == 5) { out[a] = act; continue; }     /* act < 0: not in the active list */
        int tx, ty;
        target(w->ag[a].x, w->ag[a].y, act, &tx, &ty);
        if (tx > w->W - 1 || tx < 0) act = 0;
        if (ty > w->H - 1 || ty < 0) act = 0;
        out[a] = act;
    }
}

/* cooking_world.py:206-221 check_collisions */
static void check_collisions(const World *w, const int *actions, int *out)
{
    int ex[CZO_MAX_AGENTS], ey[CZO_MAX_AGENTS], wk[CZO_MAX_AGENTS];
    for (int a = 0; a < w->A; ++a) {
        int tx, ty;
        if (actions[a] < 0) continue;
        target(w->ag[a].x, w->ag[a].y, actions[a], &tx, &ty);
        wk[a] = walkable(w, tx, ty);
        ex[a] = wk[a] ? tx : w->ag[a].x;
        ey[a] = wk[a] ? ty : w->ag[a].y;
    }
    for (int a = 0; a < w->A; ++a) {
        int clash = 0;
        if (actions[a] < 0) { out[a] = -1; continue; }
        /* only the agents that act take part (world_step passes compute_active_agents(), cooking_world.py:105,108) */
        for (int b = 0; b < w->A; ++b) if (b != a && actions[b] >= 0 && ex[b] == ex[a] && ey[b] == ey[a]) clash = 1;
        out[a] = (clash && wk[a]) ? 0 : actions[a];
    }
}

/* action_scheme3.py:4-23 / action_scheme1.py:4-20 perform_agent_actions */
static void perform_agent_actions(World *w, const int *actions)
{
    int cleaned[CZO_MAX_AGENTS], coll[CZO_MAX_AGENTS], tx[CZO_MAX_AGENTS], ty[CZO_MAX_AGENTS];
    int scheme = w->cfg->action_scheme;
    /* actions[a] < 0: agent a is despawned -- it is not in the list world_step acts on (cooking_world.py:105-108), but it
       stays in world.agents (location, orientation; it still blocks interactions aimed at its cell, :116) */
    for (int a = 0; a < w->A; ++a) {
        int act = actions[a];
        if (act >= 1 && act <= 4) {
            target(w->ag[a].x, w->ag[a].y, act, &tx[a], &ty[a]);
            w->ag[a].orient = act;                          /* change_orientation before any filtering */
        } else { tx[a] = w->ag[a].x; ty[a] = w->ag[a].y; }
    }
    check_inbounds(w, actions, cleaned);
    check_collisions(w, cleaned, coll);
    for (int a = 0; a < w->A; ++a) {
        Agent *ag = &w->ag[a];
        int act = coll[a];
        if (act < 0) continue;
        if (scheme == 3) {
            int moved = resolve_walking_action(w, ag, act);  /* runs for act == 0 too (re-presses a Switch) */
            if (!moved && act != 0) scheme3_interaction(w, ag, tx[a], ty[a]);
        } else {
            if (act >= 1 && act <= 4) resolve_walking_action(w, ag, act);
            else if (act == 5) resolve_primary_interaction(w, ag);
            else if (act == 6) resolve_pick_up_special(w, ag);
            else if (act == 7) resolve_execute_action(w, ag);
        }
    }
}

/* cooking_world.py:77-88 progress_world (+ Blender.process world_objects.py:321-335, BlenderFood.blend
   abstract_classes.py:266-273) */
static void progress_world(World *w)
{
    for (int c = 0; c < w->W * w->H; ++c) {
        Stat *st = &w->cell[c];
        if (st->type != BLENDER) continue;
        if (st->ncontent > 0 && st->toggle) {
            for (int i = 0; i < st->ncontent; ++i) {
                Dyn *o = &w->obj[st->content[i]];
                if (!done_(o)) o->mashed = 1;               /* FRESH -> (progress 1->0) -> MASHED in one call */
            }
            int all = 1;
            for (int i = 0; i < st->ncontent; ++i) if (!w->obj[st->content[i]].mashed) all = 0;
            if (all) { st->toggle = !st->toggle; st->ready = 0; }
        }
    }
    for (int c = 0; c < w->W * w->H; ++c) {
        Stat *st = &w->cell[c];
        if (st->ncontent > 0) {
            for (int i = 0; i < st->ncontent; ++i) w->obj[st->content[i]].free_ = 0;
            w->obj[st->content[st->ncontent - 1]].free_ = 1;
        }
    }
    for (int p = 0; p < w->D; ++p) {
        Dyn *pl = &w->obj[p];
        if (!pl->alive || pl->cls != PLATE || pl->ncontent == 0) continue;
        for (int i = 0; i < pl->ncontent; ++i) w->obj[pl->content[i]].free_ = 0;
        w->obj[pl->content[pl->ncontent - 1]].free_ = 1;
    }
}

/* cooking_world.py:90-92 resolve_linked_interactions; Switch.process_linked_objects world_objects.py:165-169;
   Block.switch_state :215-216.  All LinkedObjects share group None, <= 1 Switch per level. */
static void resolve_linked_interactions(World *w)
{
    for (int c = 0; c < w->W * w->H; ++c) {
        Stat *st = &w->cell[c];
        if (st->type != SWITCH) continue;
        if (st->pressed)
            for (int b = 0; b < w->W * w->H; ++b) if (w->cell[b].type == BLOCK) w->cell[b].walk = !w->cell[b].walk;
        st->pressed = 0;
    }
}

/* cooking_world.py:104-112 world_step (its last call, handle_agent_spawn, is made by czo_step_env: it needs the record's
   status word and the keys of the draws) */
static void world_step(World *w, const int *actions)
{
    perform_agent_actions(w, actions);
    progress_world(w);
    resolve_linked_interactions(w);
}

/* ------------------------------------------------------------------------------------------ */
/* recipes: recipe.py:77-104                                                                   */

typedef struct { int n; uint32_t node[CZO_MAX_NODES]; uint32_t children[CZO_MAX_NODES]; } Recipe;

/* narrow rows: n, then per node  class | cond << 8 | child mask << 16 | counts << 24
 * wide rows:   n, then per node two words:  class | cond << 8 | counts << 24,  child mask (16 bits) */
static void load_recipe(Recipe *r, const czo_config *cfg, const uint32_t *table, int id)
{
    if (cfg->recipe_nodes > 8) {
        const uint32_t *p = table + (size_t)id * (1 + 2 * CZO_MAX_NODES);
        r->n = (int)p[0];
        for (int j = 0; j < CZO_MAX_NODES; ++j) { r->node[j] = p[1 + 2 * j]; r->children[j] = p[2 + 2 * j] & 0xFFFFu; }
    } else {
        const uint32_t *p = table + (size_t)id * (1 + 8);
        r->n = (int)p[0];
        for (int j = 0; j < CZO_MAX_NODES; ++j) {
            r->node[j] = j < 8 ? p[1 + j] : 0;
            r->children[j] = j < 8 ? (p[1 + j] >> 16) & 255u : 0;
        }
    }
}

/* the marks of recipe r inside the record: 8 bits per recipe in word 1, or 16 bits per recipe in words 1 (recipes 0, 1)
 * and 7 (recipes 2, 3) */
static uint32_t get_marks(const czo_config *cfg, const uint32_t *rec, int r)
{
    if (cfg->recipe_nodes > 8) return (rec[r < 2 ? W_MARKS : W_RES1] >> (16 * (r & 1))) & 0xFFFFu;
    return (rec[W_MARKS] >> (8 * r)) & 255u;
}
static void set_marks(const czo_config *cfg, uint32_t *rec, int r, uint32_t m)
{
    if (cfg->recipe_nodes > 8) {
        uint32_t *w = &rec[r < 2 ? W_MARKS : W_RES1];
        *w = (*w & ~(0xFFFFu << (16 * (r & 1)))) | (m << (16 * (r & 1)));
    } else {
        rec[W_MARKS] = (rec[W_MARKS] & ~(255u << (8 * r))) | (m << (8 * r));
    }
}

/* returns marks bitmask (bit j = node j of node_list marked) */
static uint32_t update_recipe_state(const World *w, const Recipe *r)
{
    /* matched object locations per node */
    static __thread int mx[CZO_MAX_NODES][CZO_MAX_CELLS], my[CZO_MAX_NODES][CZO_MAX_CELLS];
    int mn[CZO_MAX_NODES];
    uint32_t marks = 0;
    for (int j = r->n - 1; j >= 0; --j) {                 /* reversed(node_list) */
        uint32_t nd = r->node[j];
        int cls = nd & 255, cond = (nd >> 8) & 255, children = (int)r->children[j];
        mn[j] = 0;
        if ((marks & (uint32_t)children) != (uint32_t)children) continue;   /* all(contains.marked) */
        /* iterate world_objects[node.name] */
        int count = (cls < 16) ? w->W * w->H : (cls < 32 ? w->D : 0);
        for (int k = 0; k < count; ++k) {
            int ox, oy, chopped = 0, mashed = 0;
            if (cls < 16) {
                if (w->cell[k].type != cls) continue;
                ox = k % w->W; oy = k / w->W;
            } else {
                const Dyn *o = &w->obj[k];
                if (!o->alive || o->cls != cls - 16) continue;
                ox = o->x; oy = o->y; chopped = o->chopped; mashed = o->mashed;
            }
            /* check_conditions recipe.py:96-104 */
            int ok = 1;
            if (cond & 0x10) {
                /* several (attr, value) conditions, handed over as the set of object states (chopped | mashed << 1)
                 * that satisfy all of them: recipe.py:96-98 loops over every condition of the node */
                ok = cls >= 16 && (((cond & 15) >> (chopped | (mashed << 1))) & 1);
            } else {
                /* blend_state exists on Carrot / Banana only (abstract_classes.py:257-264); on any other class the
                 * reference's getattr raises -- documented deviation: such a node matches nothing */
                int has_blend = cls == 16 + CARROT || cls == 16 + BANANA;
                if (cond == COND_CHOPPED) ok = chopped; else if (cond == COND_MASHED) ok = has_blend && mashed;
                else if (cond == COND_NOT_CHOPPED) ok = !chopped; else if (cond == COND_NOT_MASHED) ok = has_blend && !mashed;
            }
            if (!ok) continue;
            for (int c = 0; c < CZO_MAX_NODES && ok; ++c) {
                if (!(children & (1 << c))) continue;
                int any = 0;
                for (int i = 0; i < mn[c]; ++i) if (mx[c][i] == ox && my[c][i] == oy) any = 1;
                if (!any) ok = 0;
            }
            if (!ok) continue;
            mx[j][mn[j]] = ox; my[j][mn[j]] = oy; mn[j]++;
            marks |= 1u << j;
        }
    }
    return marks;
}

/* Recipe.goals_completed recipe.py:36-40 summed: number of goal slots that read "open" */
static int open_goals(const Recipe *r, uint32_t marks)
{
    int n = 0;
    for (int j = 0; j < r->n; ++j) if (((r->node[j] >> 24) & 1) && !(marks & (1u << j))) n++;
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* observation: cooking_env.py:352-373 get_feature_vector                                      */

static void observe_agent(const World *w, const czo_layout *lay, const czo_meta_entry *meta, int n_meta, int me, double *out)
{
    int n = 0;
    const double W = (double)w->W, H = (double)w->H;
    int ax = w->ag[me].x, ay = w->ag[me].y;
    for (int m = 0; m < n_meta; ++m) {
        int cls = meta[m].cls, num = meta[m].num, cur = 0, flen;
        if (cls < 16) {
            flen = (cls == FLOOR) ? 0 : ((cls == SWITCH || cls == BLOCK) ? 4 : 3);
            for (int i = lay->static_off[cls]; i < lay->static_off[cls + 1]; ++i) {
                int c = lay->static_cells[i];
                const Stat *st = &w->cell[c];
                if (flen == 0) { cur++; continue; }
                out[n++] = (double)(c % w->W - ax) / W;
                out[n++] = (double)(c / w->W - ay) / H;
                if (cls == SWITCH) out[n++] = st->active ? 1.0 : 0.0;
                if (cls == BLOCK) out[n++] = st->walk ? 1.0 : 0.0;
                out[n++] = 1.0;
                cur++;
            }
        } else if (cls < 32) {
            int dc = cls - 16;
            flen = (dc == PLATE) ? 3 : (is_blender_food(dc) ? 6 : 5);
            for (int s = 0; s < w->D; ++s) {
                const Dyn *o = &w->obj[s];
                if (!o->alive || o->cls != dc) continue;
                out[n++] = (double)(o->x - ax) / W;
                out[n++] = (double)(o->y - ay) / H;
                if (dc != PLATE) {
                    out[n++] = done_(o) ? 0.0 : 1.0;
                    if (is_blender_food(dc)) { out[n++] = o->chopped ? 1.0 : 0.0; out[n++] = o->mashed ? 1.0 : 0.0; }
                    else out[n++] = done_(o) ? 1.0 : 0.0;
                }
                out[n++] = 1.0;
                cur++;
            }
        } else {
            flen = 7;
            for (int a = 0; a < w->A; ++a) {
                const Agent *g = &w->ag[a];
                if (a == me) { out[n++] = (double)g->x / W; out[n++] = (double)g->y / H; }
                else { out[n++] = (double)(g->x - ax) / W; out[n++] = (double)(g->y - ay) / H; }
                for (int o = 1; o <= 4; ++o) out[n++] = g->orient == o ? 1.0 : 0.0;
                out[n++] = 1.0;
                cur++;
            }
        }
        for (int k = 0; k < (num - cur) * flen; ++k) out[n++] = 0.0;
    }
    /* n == feat_len by construction of feat_len (cooking_env.py:114-117) when counts <= meta */
}

/* ------------------------------------------------------------------------------------------ */
/* C entry points                                                                              */

typedef struct {
    const czo_config *cfg;
    const uint32_t *recipe_table;     /* [n_recipes][1 + 8] */
    const czo_layout *layouts;        /* [num_layouts] */
    const czo_meta_entry *meta;
    int n_meta;
    int64_t env_id_base;              /* global id of env 0 (shard offset) */
    int32_t pool_groups, pool_active; /* auto-reset draws come from part pool_active of every pool slice cut into pool_groups
                                         equal parts (0 or 1 groups: the whole slice) -- mirrors cz_set_layout_group */
    const struct czo_spawn *spawn;    /* agent despawn / respawn with keyed draws (NULL: off) -- mirrors cz_set_spawn */
} czo_ctx;

/* Agent despawn / respawn for a batch of worlds: the reference's rule (cooking_world.py:267-290) with every draw taken from
 * a counter-based stream instead of numpy's / Python's process-global ones.  Pinned by tests/golden/spawn_keyed_*.npz:
 * trajectories of the UNMODIFIED reference functions whose np.random.random / random.sample were fed with exactly these draws
 * (tools/gen_golden.py capture_spawn_keyed_episode). */
typedef struct czo_spawn {
    double despawn_rate, respawn_rate;
    uint64_t seed;
    int32_t grace_period;
    int32_t n_levels, stride;         /* spawn areas per level: up to `stride` candidates per list */
    const uint8_t *level_of_layout;   /* [num_layouts]; NULL: every layout is level 0 */
    const int32_t *n_x, *n_y;         /* [n_levels][num_agents] */
    const int32_t *xs, *ys;           /* [n_levels][num_agents][stride]: X_POSITION / Y_POSITION of the level file's AGENTS entries */
} czo_spawn;

/* status word: bit 8 + a = agent a is despawned (not in world.active_agents); from bit 12 world.agent_grace_period[a], `bits` wide each:
   5 bits while the grace period is at most 31, else 20 / A bits (cz_device.h spawn_grace_bits) */
enum { SPAWN_GONE0 = 8, SPAWN_GRACE0 = 12 };
static uint32_t spawn_grace_bits(int32_t grace_period, int A) { return grace_period <= 31 ? 5u : 20u / (uint32_t)A; }

/* the keyed stream: splitmix64 finaliser over (seed, global env id), (episode << 32 | t), (agent, draw index) -> [0, 1) */
static uint64_t spawn_mix(uint64_t x)
{
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
double czo_spawn_uniform(uint64_t seed, uint64_t env_global, uint64_t step_key, uint32_t agent, uint32_t draw)
{
    uint64_t k = spawn_mix(seed + 0x9E3779B97F4A7C15ull * env_global);
    k = spawn_mix(k ^ (step_key * 0xD1B54A32D192ED03ull));
    k = spawn_mix(k ^ ((uint64_t)agent << 32) ^ (uint64_t)draw);
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

/* parsing.py:154-167 generate_location: random.sample(x_positions, 1)[0], random.sample(y_positions, 1)[0] until the cell is a
   Floor nobody (active or not) stands on; up to 1001 tries.  Draw 2 + 2 k picks x, draw 3 + 2 k picks y in try k.  Returns 0 if
   no try succeeded (the reference raises ValueError; so it does for a candidate beyond the grid, which is skipped here). */
static int generate_location(const World *w, const czo_spawn *sp, int level, int agent, uint64_t env_global, uint64_t key, int *ox, int *oy)
{
    const int A = w->A;
    const int nx = sp->n_x[level * A + agent], ny = sp->n_y[level * A + agent];
    const int32_t *xs = sp->xs + ((size_t)level * A + agent) * sp->stride, *ys = sp->ys + ((size_t)level * A + agent) * sp->stride;
    for (int time_out = 0; time_out <= 1000; ++time_out) {
        int x = xs[(int)(czo_spawn_uniform(sp->seed, env_global, key, (uint32_t)agent, 2u + 2u * (uint32_t)time_out) * (double)nx)];
        int y = ys[(int)(czo_spawn_uniform(sp->seed, env_global, key, (uint32_t)agent, 3u + 2u * (uint32_t)time_out) * (double)ny)];
        if (x < 0 || y < 0 || x >= w->W || y >= w->H) continue;
        if (w->cell[y * w->W + x].type != FLOOR) continue;                  /* world.get_objects_at((x, y), Floor) */
        int taken = 0;
        for (int j = 0; j < A; ++j) if (w->ag[j].x == x && w->ag[j].y == y) taken = 1;     /* any agent, active or not */
        if (taken) continue;
        *ox = x; *oy = y;
        return 1;
    }
    return 0;
}

/* cooking_world.py:267-277 handle_agent_spawn (+ despawn_agent :279-285, respawn_agent :286-290) on the unpacked world;
   `status` = the record's status word (despawned bits, grace counters).  Returns the agents that left in this step (bit a):
   cooking_env.py:344-349 reports them truncated once. */
static uint32_t handle_agent_spawn(World *w, const czo_spawn *sp, uint32_t *status, int level, uint64_t env_global, uint64_t key)
{
    uint32_t st = *status, gone = 0;
    const uint32_t gbits = spawn_grace_bits(sp->grace_period, w->A), gmask = (1u << gbits) - 1u;
    for (int i = 0; i < w->A; ++i) {
        const uint32_t gsh = SPAWN_GRACE0 + gbits * (uint32_t)i;
        if ((st >> gsh) & gmask) { st -= 1u << gsh; continue; }           /* agent_grace_period[i] -= 1 */
        int n_active = 0;
        for (int j = 0; j < w->A; ++j) n_active += !((st >> (SPAWN_GONE0 + j)) & 1u);
        const int active = !((st >> (SPAWN_GONE0 + i)) & 1u);
        /* the `and` chain draws only when it gets that far; an `elif` arm is looked at only when the `if` test failed */
        if (n_active > 1 && active && czo_spawn_uniform(sp->seed, env_global, key, (uint32_t)i, 0u) < sp->despawn_rate) {
            if (w->ag[i].holding >= 0) continue;                           /* despawn_agent: an agent that holds something stays */
            st |= 1u << (SPAWN_GONE0 + i);
            gone |= 1u << i;
        } else if (!active && czo_spawn_uniform(sp->seed, env_global, key, (uint32_t)i, 1u) < sp->respawn_rate) {
            st &= ~(1u << (SPAWN_GONE0 + i));                              /* respawn_agent */
            st |= (uint32_t)sp->grace_period << gsh;
            int x, y;
            if (generate_location(w, sp, level, i, env_global, key, &x, &y)) { w->ag[i].x = x; w->ag[i].y = y; }   /* the location only (:290) */
        }
    }
    *status = st;
    return gone;
}
static uint32_t spawn_initial_status(const czo_spawn *sp, int A)     /* load_level.py:67-68, parsing.py:142 */
{
    uint32_t st = 0;
    for (int a = 0; a < A; ++a) st |= (uint32_t)sp->grace_period << (SPAWN_GRACE0 + spawn_grace_bits(sp->grace_period, A) * (uint32_t)a);
    return st;
}

static void recompute_marks(const czo_ctx *cx, World *w, uint32_t *rec)
{
    rec[W_MARKS] = 0;
    if (cx->cfg->recipe_nodes > 8) rec[W_RES1] = 0;
    for (int r = 0; r < cx->cfg->num_recipes; ++r) {
        int id = (rec[W_RECIPES] >> (8 * r)) & 255;
        Recipe R; load_recipe(&R, cx->cfg, cx->recipe_table, id);
        set_marks(cx->cfg, rec, r, update_recipe_state(w, &R));
    }
}

/* layout an env draws for its k-th episode; keyed by the GLOBAL env id so sharding does not change results */
uint32_t czo_next_layout(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t num_layouts)
{
    uint32_t base = pool_word & 0xFFFFu, count = pool_word >> 16;
    if (count == 0) { base = 0; count = num_layouts; }
    return base + (uint32_t)(((uint64_t)env_global + (uint64_t)episode * 7919u) % count);
}
uint32_t czo_next_layout_group(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t num_layouts, uint32_t groups, uint32_t active)
{
    uint32_t base = pool_word & 0xFFFFu, count = pool_word >> 16;
    if (count == 0) { base = 0; count = num_layouts; }
    if (groups > 1) { count /= groups; base += active * count; }
    return base + (uint32_t)(((uint64_t)env_global + (uint64_t)episode * 7919u) % count);
}

/* counter-based action stream shared by oracle, kernel and host (mirrors: cz_device.h action_hash) */
uint32_t czo_action(uint64_t seed, int64_t env_global, int agent, uint32_t step, uint32_t n_actions)
{
    /* counter-based: two rounds of a 32-bit avalanche mixer over (seed, env, agent, step) */
    uint32_t x = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u);
    x ^= (uint32_t)env_global * 0x85EBCA6Bu + (uint32_t)((uint64_t)env_global >> 32) * 0x27D4EB2Fu;
    x ^= ((uint32_t)agent + 1u) * 0xC2B2AE35u;
    x ^= (step + 1u) * 0x165667B1u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    x += step * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (uint32_t)(((uint64_t)x * (uint64_t)n_actions) >> 32);
}

/* reset one env from a layout: cooking_env.py:178-210 (t=0, marks re-evaluated, obs of the fresh world) */
int czo_reset_env(const czo_ctx *cx, int64_t env_local, uint32_t layout_id, uint32_t *rec, double *obs)
{
    const czo_config *cfg = cx->cfg;
    uint32_t recipes = rec[W_RECIPES], episode = rec[W_EPISODE], pool = rec[W_POOL];
    memcpy(rec, cx->layouts[layout_id].init_record, sizeof(uint32_t) * (size_t)cfg->record_words);
    rec[W_T] = 0; rec[W_LAYOUT] = layout_id; rec[W_STATUS] = 0; rec[W_EPISODE] = episode; rec[W_RECIPES] = recipes; rec[W_POOL] = pool;
    if (cx->spawn) rec[W_STATUS] = spawn_initial_status(cx->spawn, cfg->num_agents);
    static __thread World w;
    unpack(&w, cfg, rec);
    recompute_marks(cx, &w, rec);
    if (obs)
        for (int a = 0; a < cfg->num_agents; ++a)
            observe_agent(&w, &cx->layouts[layout_id], cx->meta, cx->n_meta, a, obs + (size_t)a * cfg->feat_len);
    return w.err;
}

/* one env, one step: cooking_env.py:243-269 accumulated_step + :290-315 compute_rewards + :333-350 + observe */
int czo_step_env(const czo_ctx *cx, int64_t env_local, uint32_t *rec, const int32_t *actions,
                 double *obs, double *rewards, uint8_t *term, uint8_t *trunc)
{
    const czo_config *cfg = cx->cfg;
    const int A = cfg->num_agents;
    static __thread World w;
    if (rec[W_STATUS] & STATUS_DONE) {
        if (cfg->auto_reset) {
            uint32_t ep = rec[W_EPISODE] + 1;
            rec[W_EPISODE] = ep;
            uint32_t lay = czo_next_layout_group(cx->env_id_base + env_local, ep, rec[W_POOL], (uint32_t)cfg->num_layouts,
                                                 (uint32_t)cx->pool_groups, (uint32_t)cx->pool_active);
            int e = czo_reset_env(cx, env_local, lay, rec, obs);
            for (int a = 0; a < A; ++a) { rewards[a] = 0.0; term[a] = 0; trunc[a] = 0; }
            return e;
        }
        unpack(&w, cfg, rec);
        for (int a = 0; a < A; ++a) {
            if (obs) observe_agent(&w, &cx->layouts[rec[W_LAYOUT]], cx->meta, cx->n_meta, a, obs + (size_t)a * cfg->feat_len);
            rewards[a] = 0.0; term[a] = !!(rec[W_STATUS] & STATUS_TERM); trunc[a] = !!(rec[W_STATUS] & STATUS_TRUNC);
        }
        return w.err;
    }
    unpack(&w, cfg, rec);
    rec[W_T] += 1;                                              /* cooking_env.py:244 */
    int acts[CZO_MAX_AGENTS];
    for (int a = 0; a < A; ++a) acts[a] = actions[a];
    uint32_t gone = 0;
    if (cx->spawn) {
        /* a despawned agent is not in the list world_step acts on (cooking_world.py:105-108) */
        for (int a = 0; a < A; ++a) if ((rec[W_STATUS] >> (SPAWN_GONE0 + a)) & 1u) acts[a] = -1;
    }
    world_step(&w, acts);                                       /* :246 (its last call, handle_agent_spawn, follows) */
    if (cx->spawn) {
        const int level = cx->spawn->level_of_layout ? cx->spawn->level_of_layout[rec[W_LAYOUT]] : 0;
        gone = handle_agent_spawn(&w, cx->spawn, &rec[W_STATUS], level, (uint64_t)(cx->env_id_base + env_local),
                                  ((uint64_t)rec[W_EPISODE] << 32) | (uint64_t)rec[W_T]);
    }

    /* compute_rewards :290-315 */
    int truncated = (int)rec[W_T] >= cfg->max_steps;            /* compute_truncated :333-350 */
    int n_completed = 0;
    double rew[CZO_MAX_RECIPES];
    for (int r = 0; r < cfg->num_recipes; ++r) {
        int id = (rec[W_RECIPES] >> (8 * r)) & 255;
        Recipe R; load_recipe(&R, cfg, cx->recipe_table, id);
        uint32_t mb = get_marks(cfg, rec, r);
        int goals_before = open_goals(&R, mb);
        int completion_before = (int)(mb & 1);                  /* root node is node_list[0] */
        uint32_t ma = update_recipe_state(&w, &R);
        set_marks(cfg, rec, r, ma);
        int goals_after = open_goals(&R, ma);
        int completed = (int)(ma & 1);
        int malus = !completed && completion_before;
        int bonus = completed && !completion_before;
        double x = 0.0;
        x += (double)(goals_before - goals_after) * cfg->recipe_node_reward;
        x += (double)bonus * cfg->recipe_reward;
        x += (double)malus * cfg->recipe_penalty;
        x += cfg->max_time_penalty / (double)cfg->max_steps;
        rew[r] = x;
        n_completed += completed;
    }
    int done = cfg->end_condition_all ? (n_completed == cfg->num_recipes) : (n_completed > 0);
    for (int a = 0; a < A; ++a) { rewards[a] = rew[a]; term[a] = (uint8_t)done; trunc[a] = (uint8_t)(truncated || ((gone >> a) & 1u)); }
    if (done || truncated)
        rec[W_STATUS] |= STATUS_DONE | (done ? STATUS_TERM : 0) | (truncated ? STATUS_TRUNC : 0);
    pack(&w, rec);
    if (obs)
        for (int a = 0; a < A; ++a)
            observe_agent(&w, &cx->layouts[rec[W_LAYOUT]], cx->meta, cx->n_meta, a, obs + (size_t)a * cfg->feat_len);
    return w.err;
}

/* batch helpers (plain loops; the CPU baseline leg of bench.py times these) */
int czo_step_batch(const czo_ctx *cx, int64_t n_envs, uint32_t *records, const int32_t *actions,
                   double *obs, double *rewards, uint8_t *term, uint8_t *trunc)
{
    const czo_config *cfg = cx->cfg;
    int err = 0;
    for (int64_t e = 0; e < n_envs; ++e) {
        int r = czo_step_env(cx, e, records + (size_t)e * cfg->record_words, actions + (size_t)e * cfg->num_agents,
                             obs ? obs + (size_t)e * cfg->num_agents * cfg->feat_len : NULL,
                             rewards + (size_t)e * cfg->num_agents, term + (size_t)e * cfg->num_agents,
                             trunc + (size_t)e * cfg->num_agents);
        if (r && !err) err = r;
    }
    return err;
}

/* T steps with on-the-fly counter-based random actions (same stream as the GPU rollout kernel);
   obs/rewards/term/trunc hold the LAST step only; step0 = global step index of the first step */
int czo_rollout(const czo_ctx *cx, int64_t n_envs, uint32_t *records, int32_t T, uint64_t seed, uint32_t step0,
                double *obs, double *rewards, uint8_t *term, uint8_t *trunc, int32_t *actions_out)
{
    const czo_config *cfg = cx->cfg;
    const int nact = cfg->action_scheme == 3 ? 5 : 8;
    int err = 0;
    for (int64_t e = 0; e < n_envs; ++e) {
        for (int32_t t = 0; t < T; ++t) {
            int32_t acts[CZO_MAX_AGENTS];
            for (int a = 0; a < cfg->num_agents; ++a) {
                acts[a] = (int32_t)czo_action(seed, cx->env_id_base + e, a, step0 + (uint32_t)t, (uint32_t)nact);
                if (actions_out) actions_out[((size_t)t * n_envs + e) * cfg->num_agents + a] = acts[a];
            }
            int r = czo_step_env(cx, e, records + (size_t)e * cfg->record_words, acts,
                                 obs ? obs + (size_t)e * cfg->num_agents * cfg->feat_len : NULL,
                                 rewards + (size_t)e * cfg->num_agents, term + (size_t)e * cfg->num_agents,
                                 trunc + (size_t)e * cfg->num_agents);
            if (r && !err) err = r;
        }
    }
    return err;
}

int czo_observe_env(const czo_ctx *cx, const uint32_t *rec, double *obs)
{
    static __thread World w;
    unpack(&w, cx->cfg, rec);
    for (int a = 0; a < cx->cfg->num_agents; ++a)
        observe_agent(&w, &cx->layouts[rec[W_LAYOUT]], cx->meta, cx->n_meta, a, obs + (size_t)a * cx->cfg->feat_len);
    return w.err;
}

int czo_sizeof_config(void) { return (int)sizeof(czo_config); }
