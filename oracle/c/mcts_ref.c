/* Oracle in plain C -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Restates oracle/search.py + oracle/ttt.py (themselves pinned to the reference by
 * tests/golden/) for sizes the Python oracle cannot finish in seconds: batches of
 * Tic-Tac-Toe self-play games with a table evaluator (post-softmax probabilities +
 * value per position).  Reference lines: Search/Explorer.py:40-210, Search/Node.py,
 * Training/Gamer.py:64-79, Games/Tic_Tac_Toe/tic_tac_toe.py.  Randomness is supplied by
 * the caller per move (numpy RandomState in oracle/cref.py), in the reference's order.
 * Build: oracle/c/build.py (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define A 9
#define T 9

typedef struct {
  int visit, to_play, n_children, child_base, action, terminal;
  double prior, value_sum;
} Node;

typedef struct {
  Node* nodes;
  int n_nodes, cap;
  int board[9];
  int length, alive, outcome, root;
  long simulations, expansions;
} Game;

typedef struct {
  int n_games, sims, training, negate_player, softmax_moves;
  double base, init, value_factor, frac, eps_softmax, eps_random;
  const float* table; /* [19683][10] */
  Game* games;
  int *hist_visits, *hist_action, *hist_tree_size, *hist_children;
  double *hist_prior, *hist_vsum, *hist_bias, *hist_root_vsum;
} Batch;

static const int LINES[8][3] = {{0,1,2},{3,4,5},{6,7,8},{0,3,6},{1,4,7},{2,5,8},{0,4,8},{2,4,6}};

static int terminal_of(const int* b, int length, int* value) { /* tic_tac_toe.py:198-262 */
  for (int p = 1; p <= 2; ++p)
    for (int l = 0; l < 8; ++l)
      if (b[LINES[l][0]] == p && b[LINES[l][1]] == p && b[LINES[l][2]] == p) { *value = p == 1 ? 1 : -1; return 1; }
  *value = 0;
  return length == 9;
}
static int code_of(const int* b) { int k = 0; for (int a = 8; a >= 0; --a) k = k * 3 + b[a]; return k; }

/* numpy pairwise sum, n <= 9 */
static double np_sum(const double* v, int n) {
  if (n < 8) { double r = 0.0; for (int i = 0; i < n; ++i) r += v[i]; return r; }
  double r = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  for (int i = 8; i < n; ++i) r += v[i];
  return r;
}
static int np_choice(const double* p, int n, double u) { /* cumsum, /= last, searchsorted right */
  double cdf[A], run = 0.0;
  for (int i = 0; i < n; ++i) { run = i == 0 ? p[0] : run + p[i]; cdf[i] = run; }
  for (int i = 0; i < n; ++i) if (cdf[i] / cdf[n - 1] > u) return i;
  return n - 1;
}

static double score(const Batch* b, const Node* parent, const Node* child) { /* Explorer.py:103-130 */
  double c = log((parent->visit + b->base + 1) / b->base) + b->init;
  double u = sqrt((double)parent->visit) / (child->visit + 1);
  double conf = child->prior * u;
  conf = conf * c;
  double q = child->visit == 0 ? 0.0 : child->value_sum / child->visit;
  if (parent->to_play == b->negate_player) q = -q;
  q = q * b->value_factor;
  return conf + q;
}

static void simulate(Batch* b, Game* g) { /* Explorer.py:49-61 */
  int path[16], plen = 0, board[9], length = g->length;
  memcpy(board, g->board, sizeof(board));
  int node = g->root;
  path[plen++] = node;
  while (g->nodes[node].n_children > 0) {
    const Node* p = &g->nodes[node];
    int best = -1; double bs = 0; int ba = -1;
    for (int j = 0; j < p->n_children; ++j) {
      const Node* c = &g->nodes[p->child_base + j];
      double s = score(b, p, c);
      if (best < 0 || s > bs || (s == bs && c->action > ba)) { best = p->child_base + j; bs = s; ba = c->action; }
    }
    board[ba] = (length % 2) + 1;
    ++length;
    node = best;
    path[plen++] = node;
  }
  Node* leaf = &g->nodes[node];
  leaf->to_play = (length % 2) + 1;
  int tv; double value;
  if (terminal_of(board, length, &tv)) {
    leaf->terminal = 1;
    value = tv;
  } else {
    const float* row = b->table + (size_t)code_of(board) * 10;
    double probs[A], mask[A];
    for (int a = 0; a < A; ++a) { mask[a] = board[a] == 0 ? 1.0 : 0.0; probs[a] = (double)row[a] * mask[a]; }
    double total = np_sum(probs, A);
    if (total == 0) { for (int a = 0; a < A; ++a) probs[a] += mask[a]; total = np_sum(probs, A); }
    leaf->child_base = g->n_nodes;
    for (int a = 0; a < A; ++a) if (mask[a] != 0.0) {
      Node* c = &g->nodes[g->n_nodes++];
      memset(c, 0, sizeof(*c));
      c->prior = probs[a] / total; c->action = a; c->to_play = -1;
      leaf = &g->nodes[node];
      leaf->n_children++;
    }
    value = (double)row[9];
    g->expansions++;
  }
  for (int i = 0; i < plen; ++i) { g->nodes[path[i]].visit += 1; g->nodes[path[i]].value_sum += value; }
  g->simulations++;
}

void* oc_create(int n_games, int sims, double base, double init, double value_factor, double frac, int softmax_moves,
                double eps_softmax, double eps_random, int training, int negate_player, const float* table) {
  Batch* b = (Batch*)calloc(1, sizeof(Batch));
  b->n_games = n_games; b->sims = sims; b->base = base; b->init = init; b->value_factor = value_factor;
  b->frac = frac; b->softmax_moves = softmax_moves; b->eps_softmax = eps_softmax; b->eps_random = eps_random;
  b->training = training; b->negate_player = negate_player; b->table = table;
  b->games = (Game*)calloc(n_games, sizeof(Game));
  for (int i = 0; i < n_games; ++i) {
    Game* g = &b->games[i];
    g->cap = 2 + sims * 45;
    g->nodes = (Node*)calloc(g->cap, sizeof(Node));
    g->nodes[0].to_play = -1;
    g->n_nodes = 1; g->alive = 1;
  }
  size_t gt = (size_t)n_games * T;
  b->hist_visits = (int*)calloc(gt * A, sizeof(int)); b->hist_action = (int*)malloc(gt * sizeof(int));
  for (size_t i = 0; i < gt; ++i) b->hist_action[i] = -1;
  b->hist_tree_size = (int*)calloc(gt, sizeof(int)); b->hist_children = (int*)calloc(gt, sizeof(int));
  b->hist_prior = (double*)calloc(gt * A, sizeof(double)); b->hist_vsum = (double*)calloc(gt * A, sizeof(double));
  b->hist_bias = (double*)calloc(gt, sizeof(double)); b->hist_root_vsum = (double*)calloc(gt, sizeof(double));
  return b;
}

void oc_destroy(void* h) {
  Batch* b = (Batch*)h;
  for (int i = 0; i < b->n_games; ++i) free(b->games[i].nodes);
  free(b->games); free(b->hist_visits); free(b->hist_action); free(b->hist_tree_size); free(b->hist_children);
  free(b->hist_prior); free(b->hist_vsum); free(b->hist_bias); free(b->hist_root_vsum); free(b);
}

void oc_state(void* h, int* alive, int* root_children) {
  Batch* b = (Batch*)h;
  for (int i = 0; i < b->n_games; ++i) {
    alive[i] = b->games[i].alive;
    root_children[i] = b->games[i].alive ? b->games[i].nodes[b->games[i].root].n_children : 0;
  }
}

/* one move for every live game: noise [G][9], uniforms [G][3] */
void oc_move(void* h, const double* noise, const double* uniforms) {
  Batch* b = (Batch*)h;
  for (int gi = 0; gi < b->n_games; ++gi) {
    Game* g = &b->games[gi];
    if (!g->alive) continue;
    Node* root = &g->nodes[g->root];
    if (b->training) /* Explorer.py:201-210 */
      for (int j = 0; j < root->n_children; ++j) {
        Node* c = &g->nodes[root->child_base + j];
        c->prior = c->prior * (1 - b->frac) + noise[(size_t)gi * A + j] * b->frac;
      }
    for (int s = 0; s < b->sims; ++s) simulate(b, g);
    root = &g->nodes[g->root];
    const int move = g->length, gm = gi * T + move, k = root->n_children;
    int counts[A], actions[A];
    for (int j = 0; j < k; ++j) {
      const Node* c = &g->nodes[root->child_base + j];
      counts[j] = c->visit; actions[j] = c->action;
      b->hist_visits[(size_t)gm * A + c->action] = c->visit;
      b->hist_prior[(size_t)gm * A + c->action] = c->prior;
      b->hist_vsum[(size_t)gm * A + c->action] = c->value_sum;
    }
    b->hist_tree_size[gm] = root->visit; b->hist_children[gm] = k;
    b->hist_bias[gm] = log((root->visit + b->base + 1) / b->base) + b->init;
    b->hist_root_vsum[gm] = root->value_sum;
    int mode = 0; double u3 = 0;
    if (b->training) { /* Explorer.py:70-97 */
      const double u1 = uniforms[gi * 3], u2 = uniforms[gi * 3 + 1]; u3 = uniforms[gi * 3 + 2];
      if (move < b->softmax_moves) mode = 1; else if (u1 < b->eps_softmax) mode = 1; else if (u2 < b->eps_random) mode = 2;
    }
    int chosen;
    if (mode == 0) { int best = 0; for (int j = 1; j < k; ++j) if (counts[j] > counts[best]) best = j; chosen = actions[best]; }
    else if (mode == 1) {
      int mx = counts[0]; for (int j = 1; j < k; ++j) if (counts[j] > mx) mx = counts[j];
      double e[A]; for (int j = 0; j < k; ++j) e[j] = exp((double)(counts[j] - mx));
      double s = np_sum(e, k); for (int j = 0; j < k; ++j) e[j] /= s;
      s = np_sum(e, k); for (int j = 0; j < k; ++j) e[j] /= s;
      chosen = actions[np_choice(e, k, u3)];
    } else {
      double m[A]; for (int a = 0; a < A; ++a) m[a] = g->board[a] == 0 ? 1.0 : 0.0;
      double n = np_sum(m, A); for (int a = 0; a < A; ++a) m[a] /= n;
      chosen = np_choice(m, A, u3);
    }
    b->hist_action[gm] = chosen;
    g->board[chosen] = (g->length % 2) + 1;
    g->length++;
    for (int j = 0; j < k; ++j) if (actions[j] == chosen) g->root = root->child_base + j;
    int tv;
    if (terminal_of(g->board, g->length, &tv)) { g->alive = 0; g->outcome = tv; }
  }
}

void oc_export(void* h, int* visits, int* actions, int* lengths, int* outcomes, int* tree_size, int* n_children,
               double* bias, double* prior, double* vsum, double* root_vsum, long* counters) {
  Batch* b = (Batch*)h;
  size_t gt = (size_t)b->n_games * T;
  memcpy(visits, b->hist_visits, gt * A * sizeof(int)); memcpy(actions, b->hist_action, gt * sizeof(int));
  memcpy(tree_size, b->hist_tree_size, gt * sizeof(int)); memcpy(n_children, b->hist_children, gt * sizeof(int));
  memcpy(bias, b->hist_bias, gt * sizeof(double)); memcpy(prior, b->hist_prior, gt * A * sizeof(double));
  memcpy(vsum, b->hist_vsum, gt * A * sizeof(double)); memcpy(root_vsum, b->hist_root_vsum, gt * sizeof(double));
  counters[0] = counters[1] = 0;
  for (int i = 0; i < b->n_games; ++i) {
    lengths[i] = b->games[i].length; outcomes[i] = b->games[i].outcome;
    counters[0] += b->games[i].simulations; counters[1] += b->games[i].expansions;
  }
}
