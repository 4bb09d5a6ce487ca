/*
 * cookingzoo.h -- C-ABI of the MI355X-native CookingZoo step() path (libcookingzoo_hip.so).
 *
 * The reference (DavidRother/cooking_zoo) has no FFI: its boundary is the Python API
 * (`cooking_zoo/environment/cooking_env.py:26-46` parallel_env, `:243` accumulated_step, `:271` observe,
 * `:178` reset).  This header is the boundary the build introduces UNDER that API: plain pointers and
 * sizes, no torch types.  Each entry point names the reference code it replaces.  The Python host
 * (cooking_zoo_amd/_native.py) binds it with ctypes; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: every function returns 0 on success, non-zero on failure with a message available from
 * cz_last_error(handle) (or cz_last_error(NULL) for creation failures).  One host thread per handle;
 * one HIP stream per handle (its own, or the caller's after cz_set_stream); the caller owns every buffer it passes; the library owns device state
 * until cz_destroy.  "d_" parameters are device pointers (from cz_dev_alloc or any HIP allocation).
 *
 * Per-env state travels as a flat "record" of cz_record_words() little-endian uint32 words:
 *   word 0 t | 1 recipe-node marks (bit 8r+j) | 2 layout id | 3 status (1 done, 2 terminated, 4 truncated)
 *   | 4 episode | 5 recipe ids (4 x u8) | 6 layout-pool slice (base | count<<16) | 7 marks of recipes 2, 3 (wide recipe tables only) | 8..11 agents (x | y<<8 | orientation<<16 | (held slot+1)<<24)
 *   | 12..19 four float64 running episode returns (statistics only) | cells (W*H bytes: type | READY<<3 | TOGGLE<<4 | ACTIVE<<5 | WALK<<6) | dyn0[D] (x | y<<8 | class<<16 | flags<<24;
 *   flags: 1 alive, 2 chopped, 4 mashed, 8 free) | dyn1[D] ((plate slot+1) | seq<<8), padded to 16 words.
 * (normative description: cooking_zoo_amd/soa.py)
 *
 * WHERE THE REFERENCE RAISES, the step kernels do something defined instead (profiles/r05/deviations.md, generated from the
 * differential fuzz's histograms by tools/deviations.py):
 *   - EXECUTE in front of a READY Cutboard with empty content (TypeError, cooking_world.py:162 <- world_objects.py:250-269; reachable
 *     in scheme1: a mashed Banana on the board is absorbed by a Plate in hand): no-op, the board stays READY
 *     (tests/golden/refcrash_cutboard_*.npz hold the reference's own outputs past that step);
 *   - a scheme1 interaction aimed off the grid (IndexError): no-op;
 *   - max_steps reached while an agent is despawned (IndexError, cooking_env.py:337): the step truncates;
 *   - a respawn that finds no free cell in 1001 tries, or a candidate beyond the grid (ValueError, parsing.py:159,166): the agent stays
 *     where it is and cz_spawn_exhausted counts it.
 * Refused before anything runs: a level with two Switches, action scheme 2, more than 4 agents (the reference's own limits or crashes).
 */
#ifndef COOKINGZOO_H
#define COOKINGZOO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The one place the ABI number lives: cz_abi_version() returns it, cooking_zoo_amd/_native.py parses it from this file
 * and refuses a library that reports another one, __graft_entry__.build() and the tests compare against it. */
#define CZ_ABI_VERSION 8

typedef struct cz_handle_s *cz_handle;

/* Batch configuration.  Mirrors the kwargs of CookingEnvironment.__init__ (cooking_env.py:62-64) that
 * reach the step path, plus the batch geometry. */
typedef struct cz_config {
    int32_t num_envs;            /* env instances owned by this handle (one wavefront each)          */
    int32_t num_agents;          /* A <= 4            (cooking_env.py:76, COLORS cooking_world.py:21) */
    int32_t width, height;       /* grid, W <= 32, H <= 32 (parsing.py:17-18)                         */
    int32_t max_dyn;             /* D dynamic-object slots per env, <= 255                            */
    int32_t feat_len;            /* F features per agent (cooking_env.py:114-117)                     */
    int32_t action_scheme;       /* 1 or 3            (cooking_env.py:60; scheme2 is dead upstream)   */
    int32_t max_steps;           /* truncation horizon (cooking_env.py:333-334)                       */
    int32_t end_condition_all;   /* end_condition_all_dishes (cooking_env.py:310-313)                 */
    int32_t num_recipes;         /* recipe graphs per env, num_agents <= R <= 4 (cooking_env.py:106)  */
    int32_t auto_reset;          /* 0: finished envs freeze until cz_reset; 1: the step after `done`
                                    re-instantiates the env from the layout pool (next-step autoreset) */
    int32_t device_id;           /* HIP device ordinal                                                */
    int64_t env_id_base;         /* global id of env 0 of this handle (shard offset; keys layout draws
                                    and the on-device action stream so results do not depend on sharding) */
    double recipe_reward, max_time_penalty, recipe_penalty, recipe_node_reward;   /* cooking_env.py:79-80 */
} cz_config;

/* Episode statistics of one handle (one GPU): what the reference reports through `infos`
 * (cooking_env.py:248,264,329), aggregated on the device. */
typedef struct cz_stats {
    uint64_t env_steps;              /* world steps executed (reset passes excluded)   */
    uint64_t episodes;               /* episodes finished                              */
    uint64_t length_sum;             /* sum of finished-episode lengths                */
    uint64_t truncations;            /* episodes that ended by t >= max_steps          */
    uint64_t terminations;           /* episodes that ended by recipe completion       */
    uint64_t recipes_completed[4];   /* per agent slot: episodes ending with its recipe complete */
    double return_sum[4];            /* per agent slot: sum of finished-episode returns */
} cz_stats;

/* ---- lifetime ------------------------------------------------------------------------------------- */
int cz_create(const cz_config *cfg, cz_handle *out);
int cz_destroy(cz_handle h);
const char *cz_last_error(cz_handle h);
int32_t cz_record_words(cz_handle h);
int32_t cz_abi_version(void);
int32_t cz_sizeof_config(void);                           /* sizeof(cz_config), for binding self-checks */
int32_t cz_sizeof_stats(void);
int cz_debug_set_stamps(cz_handle h, void *d_buf);       /* diagnostic builds only (make prof): phase stamp buffer */
/* timeline builds only (make timeline, tools/timeline.py): every step-kernel launch k writes two uint64 per env - entry and
 * exit time of the env's wave on the device-wide 100 MHz clock, with the hardware ids in the upper halves - to
 * d_buf + (k % cap_launches) * num_envs * 2 */
int cz_debug_set_timeline(cz_handle h, void *d_buf, int32_t cap_launches);
int cz_sync(cz_handle h);                                  /* wait for the handle's stream */
/* Order all further work of this handle on a HIP stream of the caller (hipStream_t; e.g. the stream a framework runs its
 * policy on: device-pointer steps then need no host synchronisation on either side).  NULL restores the handle's own
 * stream.  The stream must belong to the handle's device and outlive its use here. */
int cz_set_stream(cz_handle h, void *hip_stream);
/* STREAM CAPTURE.  While that stream is being captured by the caller (hipStreamBeginCapture, torch.cuda.graph), the
 * device-pointer calls - cz_step_device, cz_step_device_compact, cz_step_device_many / _ring, cz_rollout*, cz_observe_device,
 * cz_probe_policy - are pure kernel launches and legal inside the capture: nothing is queried or synchronised (a layout update
 * staged by cz_update_layouts stays staged until the first call outside the capture; ring runs go out as plain launches).  Replays
 * of the caller's graph then do exactly what the captured launches did (cooking_env.py:243-288 once per captured step).  Calls
 * that copy to / from the host or wait (cz_step, cz_reset, cz_get_state, cz_sync, cz_get_stats, cz_update_layouts,
 * cz_set_layout_group ...) are not; the last two say so, the others fail with HIP's own error.  The captured launches carry the
 * table pointers of their time: after cz_load_layouts / cz_load_recipes / cz_set_spawn / cz_set_compact_output capture again.
 * cz_stream_capturing: 1 while the handle's stream is a stream of the caller that is being captured, else 0 - a host layer that
 * keeps count of the steps it issued (cooking_zoo_amd: layout rotation schedules) must not count captured launches as steps:
 * they run when the caller's graph is replayed, as often as it is replayed. */
int32_t cz_stream_capturing(cz_handle h);

/* ---- tables ---------------------------------------------------------------------------------------- */
/* Recipe graphs: replaces RECIPES[name]() / Recipe.node_list (recipe_drawer.py:109-118, recipe.py:29-34).
 * max_nodes = 8 (compact): table[n][9], word 0 = node count (<= 8), then per node (root first, node_list order)
 *   class | condition<<8 | child-mask<<16 | counts-in-goal-sum<<24; the marks of recipe r are bits 8r.. of record word 1.
 * max_nodes = 16 (wide, for books with a graph of more than 8 nodes): table[n][33], word 0 = node count (<= 16), then per
 *   node TWO words: class | condition<<8 | counts<<24, child-mask (16 bits); marks are 16 bits per recipe: recipes 0, 1 in
 *   record word 1, recipes 2, 3 in word 7.  Wide batches re-evaluate every recipe on every step that changed an object.
 * class: 0..6 static type, 16..25 dynamic class, 255 none; condition: 0 none, 1 chopped, 2 mashed, 3 not chopped,
 * 4 not mashed, or 0x10 | accept mask over the object state (chopped | mashed << 1) for several (attr, value) conditions
 * on one node (recipe.py:96-98 loops over all of them). */
int cz_load_recipes(cz_handle h, const uint32_t *table, int32_t n_recipes, int32_t max_nodes);

/* Layout pool: replaces load_level / parsing (load_level.py:55-71, parsing.py:5-151) results.  Per layout:
 * its initial record (cz_record_words words; t, marks, status ignored) and its observation descriptor
 * (feat_len words, op | ref<<8; ops listed in cooking_zoo_amd/soa.py) which encodes the meta-file class
 * order and per-class list order that get_feature_vector (cooking_env.py:352-373) iterates. */
int cz_load_layouts(cz_handle h, const uint32_t *init_records, const uint32_t *obs_desc, int32_t n_layouts);

/* Fresh layouts under a stepping batch (the reference draws a new level at every reset, cooking_env.py:191-195,
 * parsing.py:21-151; the device redraws from the resident pool).  cz_update_layouts replaces pool slots
 * [first, first + count) - same formats as cz_load_layouts - without touching the handle's stream: the tables stay where they
 * are, the caller's arrays are free again on return, and the copy runs on a stream of the library's own, issued once every
 * step issued before the call has completed (checked at later call boundaries, never a device-side wait) and not waited for
 * by later steps.  Never replace slots that envs can still draw or are still
 * playing on: cz_set_layout_group(h, groups, active) cuts every env's pool slice into `groups` equal parts and lets the envs
 * draw their next episodes from part `active` only (1, 0 = the whole slice, the default), from the next step on (stream
 * order); it waits - on the device - for the updates issued so far, so no env draws a half-written slot.  Rotation:
 * switch to part b; issue at least max_steps + 1 more steps (every episode that started on part a has ended); update
 * part a; ... (cooking_zoo_amd.vec_env.CookingVecEnv.rotate_layouts does this with a background process).  Every pool slice's
 * length must be a multiple of `groups`.  cz_update_layouts with count 0 only creates the copy stream and the staging block
 * (so that the first real update does not pay for them).  cz_layout_updates: slots replaced so far. */
int cz_update_layouts(cz_handle h, int32_t first, int32_t count, const uint32_t *init_records, const uint32_t *obs_desc);
int cz_set_layout_group(cz_handle h, int32_t groups, int32_t active);
int64_t cz_layout_updates(cz_handle h);

/* Agent despawn / respawn (cooking_world.py:267-290 handle_agent_spawn, despawn_agent, respawn_agent; parsing.py:154-167
 * generate_location) for every world of the batch, evaluated by the step kernels themselves - on every path: cz_step,
 * cz_step_device*, ring runs, cz_rollout, cz_rollout_actions.  The reference takes its draws from numpy's process-global
 * stream, which defines them for one world per process; here every draw comes from a counter-based stream keyed by (seed,
 * global env id, episode << 32 | t, agent, draw index) (cz_spawn_uniform is the host mirror), so results depend neither on the
 * batch size nor on the sharding nor on the launch form; tests/golden/spawn_keyed_*.npz are trajectories of the unmodified
 * reference functions fed with exactly these draws.  The record's status word carries a "despawned" bit per agent (bit 8 +
 * agent) and a grace countdown each from bit 12 on: 5 bits wide while grace_period <= 31, else 20 / num_agents bits (so grace_period may be
 * up to 1023 for two agents, 63 for three, 31 for four; the reference takes any int, parsing.py:142).  A despawned agent does not act (whatever its action
 * says), is reported truncated in the step it leaves, and stays in the world as an obstacle; an agent that holds something stays.
 * TAKES EFFECT WITH THE NEXT STEP: worlds reset from then on start with everybody present and the grace period running
 * (parsing.py:142); episodes that are already running continue with whatever their status words hold (grace 0 unless the
 * caller set it), i.e. they draw from their next step on.  A call that moves grace_period across 31 changes the width of the
 * countdown fields: the resident records are re-packed to the new width by the call itself (a countdown that does not fit the
 * narrower field becomes 31).  A record does not say which width it was packed with: cz_set_state takes records in the width
 * of the handle's CURRENT grace_period (records saved under the other width must be re-packed by the caller,
 * cooking_zoo_amd/spawn.py decode_status / status_bits).  Rates 0, 0 switch it off.
 * Spawn areas (the level files' AGENTS entries, parsing.py:118-151) are per level: level_of_layout[i] (NULL: all 0) is the
 * level that layout i of the resident pool instantiates - call again after cz_load_layouts changed the pool size -,
 * spawn_x / spawn_y are [n_levels][num_agents][stride] candidate coordinates, n_x / n_y [n_levels][num_agents] how many of
 * them count (1..stride, stride <= 1024).  Differences from the reference, which raises in both cases: a candidate outside the
 * grid is skipped, and a respawn that finds no free Floor cell in 1001 tries puts the agent back where it stood -
 * cz_spawn_exhausted counts those. */
int cz_set_spawn(cz_handle h, double despawn_rate, double respawn_rate, int32_t grace_period, uint64_t seed,
                 int32_t n_levels, const uint8_t *level_of_layout, int32_t stride, const uint8_t *spawn_x, const int32_t *n_x,
                 const uint8_t *spawn_y, const int32_t *n_y);
int64_t cz_spawn_exhausted(cz_handle h);
double cz_spawn_uniform(uint64_t seed, int64_t env_global, uint32_t episode, uint32_t t, int32_t agent, uint32_t draw);

/* ---- state ----------------------------------------------------------------------------------------- */
int cz_set_state(cz_handle h, int64_t env_begin, int64_t env_count, const uint32_t *records);
int cz_get_state(cz_handle h, int64_t env_begin, int64_t env_count, uint32_t *records);

/* reset(): cooking_env.py:178-210.  Re-instantiates envs [env_begin, env_begin+env_count) from
 * layout_ids[i], assigns recipe_ids[i][0..3] (0xFF = unused) and pool_words[i] (base | count<<16: the slice of
 * the layout pool the env redraws from under auto_reset; NULL = whole pool), evaluates the recipe marks of the
 * fresh world (cooking_env.py:197-198) and, if obs != NULL (host, [env_count][A][F] float64), returns observe(). */
int cz_reset(cz_handle h, int64_t env_begin, int64_t env_count, const int32_t *layout_ids,
             const uint8_t *recipe_ids, const uint32_t *pool_words, double *obs);

/* observe() (cooking_env.py:271,352-373) of the current state of envs [env_begin, env_begin+env_count):
 * host buffer [env_count][A][F] float64. */
int cz_observe(cz_handle h, int64_t env_begin, int64_t env_count, double *obs);
/* ... the same observation in its compact form (see cz_step_device_compact): host buffer uint8 [env_count][A][cz_codes_pitch]. */
int cz_observe_compact(cz_handle h, int64_t env_begin, int64_t env_count, uint8_t *codes);
/* ... and into DEVICE buffers, float64 rows [env_count][A][F] and / or codes [env_count][A][cz_codes_pitch] (either may be NULL):
 * the first observation of a consumer that stays on the device, after cz_reset / cz_set_state.  Stream-ordered, nothing is
 * synchronised. */
int cz_observe_device(cz_handle h, int64_t env_begin, int64_t env_count, double *d_obs, uint8_t *d_codes);

/* ---- step ------------------------------------------------------------------------------------------ */
/* One batched env step = accumulated_step (cooking_env.py:243-269): world_step (cooking_world.py:104-112),
 * compute_rewards (cooking_env.py:290-315), compute_truncated (:333-350), then observe (:271, :352-373).
 * An action < 0 marks an agent that is not in the list world_step acts on (despawned, cooking_world.py:105-108,267-290):
 * it does not turn, move, interact or take part in the collision filter, but keeps its cell.
 * Host-pointer form: actions int32 [N][A]; outputs obs float64 [N][A][F] (may be NULL), rewards float64 [N][A],
 * terminations / truncations uint8 [N][A].  Synchronous. */
int cz_step(cz_handle h, const int32_t *actions, double *obs, double *rewards, uint8_t *terminations,
            uint8_t *truncations);

/* Recipe-node marks (record word 1: bit 8r+j = node j of the env's r-th recipe is met, bit 8r = recipe r complete) of
 * every env after the most recent cz_step: what compute_infos reports as `recipe_done` (cooking_env.py:317-331).
 * uint32 [N][2]: record words 1 and 7 (the second is 0 unless the recipe tables are wide).  Fails if cz_step has not run
 * on this handle. */
int cz_last_marks(cz_handle h, uint32_t *marks);

/* Device-pointer form, asynchronous on the handle's stream (d_obs may be NULL: encode skipped). */
int cz_step_device(cz_handle h, const int32_t *d_actions, double *d_obs, double *d_rewards,
                   uint8_t *d_terminations, uint8_t *d_truncations);

/* The same step with a COMPACT observation for consumers that stay on the device: every feature of get_feature_vector
 * (cooking_env.py:352-373) is one of at most 256 values - quotients (x - ax) / W, (y - ay) / H, 0.0, 1.0 - so d_codes
 * (uint8 [N][A][cz_codes_pitch(h)], pitch = F rounded up to 16, padding bytes 255) receives for every feature the INDEX of its
 * value in a table of 256 float64 (cz_obs_table: a host copy; cz_obs_table_device: the resident one), i.e.
 * table[d_codes[e][a][f]] is bit for bit the float64 observation - 1/8 of the bytes, losslessly.  d_obs may be NULL (codes
 * only) or a float64 [N][A][F] buffer that is written as well. */
int cz_step_device_compact(cz_handle h, const int32_t *d_actions, uint8_t *d_codes, double *d_obs, double *d_rewards,
                           uint8_t *d_terminations, uint8_t *d_truncations);
/* ... or as a setting of the handle: from now on every one-step launch (cz_step_device, _many, _ring, cz_step) writes the compact
 * observation to d_codes as well - with d_obs = NULL in those calls, instead of the float64 rows; NULL switches it off. */
int cz_set_compact_output(cz_handle h, uint8_t *d_codes);
/* host-pointer form (synchronous, like cz_step) that returns the codes instead of the float64 rows: 1/8 of the bytes over PCIe */
int cz_step_compact(cz_handle h, const int32_t *actions, uint8_t *codes, double *rewards, uint8_t *terminations, uint8_t *truncations);
int32_t cz_codes_pitch(cz_handle h);
int cz_obs_table(cz_handle h, double table[256]);
const void *cz_obs_table_device(cz_handle h);

/* K consecutive steps, one launch each, issued from C: step k reads its actions at d_actions + (k % action_period) *
 * action_stride (in int32 elements) and overwrites the same output buffers.  Equivalent to K cz_step_device calls. */
int cz_step_device_many(cz_handle h, int32_t K, const int32_t *d_actions, int64_t action_stride, int32_t action_period,
                        double *d_obs, double *d_rewards, uint8_t *d_terminations, uint8_t *d_truncations);

/* The same for actions kept in a ring of `action_period` slots (slot s at d_ring + s * action_stride): step k reads slot
 * (first_slot + k) % action_period.  Runs of consecutive slots (cut where the ring wraps and at 256 launches; at least
 * 48 launches long - CZ_GRAPH_MIN_RUN) are captured into HIP graphs on first use, keyed by (first slot, length), and replayed
 * afterwards (no host work per launch, stable kernel-argument memory); shorter pieces are launched directly (a graph's first
 * kernel starts later than a plain launch's, which a short run cannot win back); identical results.
 * At most 64 graphs are kept per handle: keep the runs of a loop aligned to the same slots. */
int cz_step_device_ring(cz_handle h, int32_t K, const int32_t *d_ring, int64_t action_stride, int32_t action_period,
                        int32_t first_slot, double *d_obs, double *d_rewards, uint8_t *d_terminations, uint8_t *d_truncations);

/* Optional: build the graphs of one such call up front (nothing is stepped), e.g. before a timed region. */
int cz_ring_prepare(cz_handle h, int32_t K, const int32_t *d_ring, int64_t action_stride, int32_t action_period,
                    int32_t first_slot, double *d_obs, double *d_rewards, uint8_t *d_terminations, uint8_t *d_truncations);

/* FUSED RING RUNS (opt-in).  With cz_set_ring_fused(h, 1), a run of two or more steps of cz_step_device_ring / _many whose action
 * slots are densely packed (action_stride == num_envs * num_agents) goes out as ONE launch per stretch of consecutive slots
 * (cut where the ring wraps): cz_rollout_actions' kernel over the ring's own rows, the env state in registers across the
 * steps, every step's observation / rewards / flags written IN PLACE to the [N][A]... buffers of the call.  Afterwards state,
 * output buffers and statistics are bit for bit what K one-step launches leave (cooking_env.py:243-269 K times) - at the cost
 * per step of a fused rollout, without launch boundaries.  Open loop by nature (the actions
 * of the whole run are read by one launch).  Runs with other strides, a compact
 * output (cz_set_compact_output) or kernel timing switched on are issued as before.  Returns the previous setting, -1 for a
 * null handle; cz_ring_fused_steps: env steps issued this way so far (reset != 0: zero the count after reading). */
int cz_set_ring_fused(cz_handle h, int32_t enabled);
int64_t cz_ring_fused_steps(cz_handle h, int32_t reset);

/* How many step kernels cz_step_device_ring has replayed from graphs / launched directly on this handle so far
 * (reset != 0: zero both after reading).  bench.py describes its run from these numbers.  Fused ring runs (cz_set_ring_fused) are
 * not in either count: cz_ring_fused_steps has theirs. */
int cz_launch_counts(cz_handle h, int64_t *graph_kernels, int64_t *direct_kernels, int32_t reset);

/* T fused steps in one launch with on-device uniform random actions (counter-based stream keyed by
 * (seed, global env id, agent, step0 + t)); state stays in registers between steps.  d_obs, if not NULL,
 * is a trajectory buffer [T][N][A][F]; d_rewards [T][N][A]; d_term/d_trunc [T][N][A] (each may be NULL
 * to keep only device-side statistics).  Asynchronous. */
int cz_rollout(cz_handle h, int32_t T, uint64_t seed, uint32_t step0, double *d_obs, double *d_rewards,
               uint8_t *d_terminations, uint8_t *d_truncations);

/* ... with the trajectory as compact observations: d_codes uint8 [T][N][A][cz_codes_pitch(h)] (cz_step_device_compact explains the
 * codes); d_obs (float64 [T][N][A][F]) may be NULL - codes only - or is written as well. */
int cz_rollout_compact(cz_handle h, int32_t T, uint64_t seed, uint32_t step0, uint8_t *d_codes, double *d_obs, double *d_rewards,
                       uint8_t *d_terminations, uint8_t *d_truncations);

/* The same fused launch over actions of the caller (replay, open-loop search): d_actions int32 [T][N][A], step t reads row t
 * (values as for cz_step_device: action & 7, negative = the agent does not act).  Same results as T cz_step_device launches
 * over those rows, with every step's outputs in the trajectory buffers. */
int cz_rollout_actions(cz_handle h, int32_t T, const int32_t *d_actions, double *d_obs, double *d_rewards,
                       uint8_t *d_terminations, uint8_t *d_truncations);

/* the action the on-device stream draws (host mirror, for parity tests) */
uint32_t cz_action(uint64_t seed, int64_t env_global, int32_t agent, uint32_t step, uint32_t n_actions);
/* the layout an env draws for its k-th episode under auto_reset */
uint32_t cz_next_layout(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t n_layouts);
/* ... when the envs draw from part `active` of `groups` (cz_set_layout_group) */
uint32_t cz_next_layout_group(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t n_layouts, uint32_t groups,
                              uint32_t active);

/* ---- device memory + timing helpers (so the Python host needs no torch) ----------------------------- */
void *cz_dev_alloc(cz_handle h, size_t bytes);
int cz_dev_free(cz_handle h, void *d_ptr);
/* pinned host memory: host buffers taken from here (actions / observations / rewards / flags of cz_step, cz_reset,
 * cz_observe) are written and read by the copy engines directly -- no per-call pinning or staging of pageable pages */
void *cz_host_alloc(cz_handle h, size_t bytes);
int cz_host_free(cz_handle h, void *ptr);
int cz_memcpy_h2d(cz_handle h, void *d_dst, const void *src, size_t bytes);
int cz_memcpy_d2h(cz_handle h, void *dst, const void *d_src, size_t bytes);
int cz_timer_start(cz_handle h);                           /* hipEventRecord on the handle's stream */
int cz_timer_stop(cz_handle h, float *elapsed_ms);         /* record + synchronize + elapsed        */
/* brackets every kernel launch of the step path with HIP events and accumulates device time */
int cz_kernel_time_reset(cz_handle h, int32_t enable);
int cz_kernel_time_read(cz_handle h, double *total_ms, int64_t *launches);
/* measurement aid: average duration of `reps` back-to-back launches of a kernel of the step kernel's grid shape that does
 * nothing but write `bytes` (<= 2 GiB) to d_dst with the encode's 16-byte write-through stores -- the floor of a launch
 * that has to emit that much output (bench.py reports it next to the roofline).  d_dst is overwritten. */
int cz_probe_output_only(cz_handle h, void *d_dst, size_t bytes, int32_t reps, float *us_per_launch);
/* measurement aid: a CLOSED loop - K times (cz_step_device, then a tiny policy kernel that derives every agent's next action
 * from the observation it was just given, same stream), one HIP graph, replayed `reps` times; average microseconds per
 * step (step + policy).  d_actions (int32 [N][A]) is read and rewritten in place; K <= 1024. */
int cz_probe_closed_loop(cz_handle h, int32_t K, int32_t reps, int32_t *d_actions, double *d_obs, double *d_rewards,
                         uint8_t *d_terminations, uint8_t *d_truncations, float *us_per_step);
/* ... the same loop over the compact observation: cz_step_device_compact (codes only), then a policy kernel that reads the same
 * four features as table indices - it takes the same actions as cz_probe_closed_loop's. */
int cz_probe_closed_loop_compact(cz_handle h, int32_t K, int32_t reps, int32_t *d_actions, uint8_t *d_codes, double *d_rewards,
                                 uint8_t *d_terminations, uint8_t *d_truncations, float *us_per_step);
/* test / measurement aid: one launch, on the handle's stream, of the stand-in policy those loops use - d_actions[env][agent] = a
 * hash of four features of the observation in d_obs (float64 rows) or, with d_obs NULL, in d_codes (compact rows; the same
 * actions).  tests/test_gpu_capture.py captures [cz_probe_policy, cz_step_device] into a graph of the caller with it. */
int cz_probe_policy(cz_handle h, const double *d_obs, const uint8_t *d_codes, int32_t *d_actions);

/* ---- statistics + multi-GPU ------------------------------------------------------------------------- */
int cz_get_stats(cz_handle h, cz_stats *out);              /* device reduction over this handle's envs */
int cz_reset_stats(cz_handle h);
/* RCCL over xGMI: the only collective of the path is an all-gather of one cz_stats per rank. */
int cz_comm_unique_id(uint8_t id[128]);
int cz_comm_init(cz_handle h, int32_t n_ranks, int32_t rank, const uint8_t id[128]);
int cz_stats_allgather(cz_handle h, cz_stats *out /* [n_ranks] */);
/* barrier over the communicator (a 4-byte all-reduce on the handle's stream + stream synchronisation) */
int cz_comm_barrier(cz_handle h);
/* diagnostics: the files that serve this process -- the RCCL library the communicator binds (a copy already loaded in
 * the process wins) and the HIP runtime this library runs on; each buffer holds `cap` bytes, either may be NULL */
int cz_runtime_paths(char *rccl_path, char *hip_path, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* COOKINGZOO_H */
