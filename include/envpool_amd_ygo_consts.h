/*
 * envpool_amd_ygo_consts.h — every ygopro-core value the YGOPro observation-encode path assumes.
 *
 * The reference takes these from ygopro-core's common.h (Fluorohydride/ygopro-core, a git submodule
 * that is EMPTY in the reference checkout and whose version is unpinned, .gitmodules:17-29). The
 * values below are the widely known upstream ones, written from memory of that header: nothing in
 * this repository can check them ("parity unpinned"). They are kept in this ONE place:
 * the device kernels (envpool_amd/csrc/ygo_kernels.hpp), the host emitter (ygo.hip) and the CPU
 * oracle (oracle/ygo_oracle.c) all take them from here, and an integrator who has the real
 * common.h checks them before the first duel:
 *
 *     ep_ygo_core_constants c;
 *     c.location[0] = LOCATION_DECK; ... c.msg[13] = MSG_ANNOUNCE_ATTRIB; ...   // from common.h
 *     if (ep_ygo_check_constants(&c) != EP_OK) { puts(ep_last_error()); abort(); }
 *
 * (ep_ygo_default_constants fills the struct with the library's own values; INTEGRATION.md shows
 * the whole stub.) A mismatch would otherwise be silent: observations of the wrong id, not a crash.
 *
 * What the reference derives from them (envpool2/ygopro/ygopro.h:531-686): id tables built by
 * make_ids over std::map, i.e. ids in ASCENDING KEY ORDER. The kernels compute those ids
 * arithmetically (location2id = bit index + 1, attribute2id = bit index + 1, race2id = bit index
 * + 1, phase2id = bit index, the 25 type columns = the set TYPE bits in ascending order), which is
 * only the reference's table if the keys are exactly the consecutive powers of two listed here:
 * ep_ygo_check_constants checks the values, the static assertions below check the structure.
 */
#ifndef ENVPOOL_AMD_YGO_CONSTS_H_
#define ENVPOOL_AMD_YGO_CONSTS_H_

#include <stdint.h>

/* LOCATION_* (ygopro.h:531-542 location2id, :2060-2137 get_cards_in_location) */
#define EP_YGO_LOCATION_DECK 0x01
#define EP_YGO_LOCATION_HAND 0x02
#define EP_YGO_LOCATION_MZONE 0x04
#define EP_YGO_LOCATION_SZONE 0x08
#define EP_YGO_LOCATION_GRAVE 0x10
#define EP_YGO_LOCATION_REMOVED 0x20
#define EP_YGO_LOCATION_EXTRA 0x40
#define EP_YGO_LOCATION_OVERLAY 0x80

/* POS_* (ygopro.h:546-559 position2id; keys in map order 0x0,0x1,0x2,0x3,0x4,0x5,0x8,0xa,0xc -> 0..8) */
#define EP_YGO_POS_FACEUP_ATTACK 0x1
#define EP_YGO_POS_FACEDOWN_ATTACK 0x2
#define EP_YGO_POS_FACEUP_DEFENSE 0x4
#define EP_YGO_POS_FACEDOWN_DEFENSE 0x8
#define EP_YGO_POS_FACEUP 0x5
#define EP_YGO_POS_FACEDOWN 0xa
#define EP_YGO_POS_ATTACK 0x3
#define EP_YGO_POS_DEFENSE 0xc

/* TYPE_* in the key order of type2str (ygopro.h:608-634): the 25 type columns of obs:cards_ */
#define EP_YGO_TYPE_MONSTER 0x1u
#define EP_YGO_TYPE_SPELL 0x2u
#define EP_YGO_TYPE_TRAP 0x4u
#define EP_YGO_TYPE_NORMAL 0x10u
#define EP_YGO_TYPE_EFFECT 0x20u
#define EP_YGO_TYPE_FUSION 0x40u
#define EP_YGO_TYPE_RITUAL 0x80u
#define EP_YGO_TYPE_TRAPMONSTER 0x100u
#define EP_YGO_TYPE_SPIRIT 0x200u
#define EP_YGO_TYPE_UNION 0x400u
#define EP_YGO_TYPE_DUAL 0x800u
#define EP_YGO_TYPE_TUNER 0x1000u
#define EP_YGO_TYPE_SYNCHRO 0x2000u
#define EP_YGO_TYPE_TOKEN 0x4000u
#define EP_YGO_TYPE_QUICKPLAY 0x10000u
#define EP_YGO_TYPE_CONTINUOUS 0x20000u
#define EP_YGO_TYPE_EQUIP 0x40000u
#define EP_YGO_TYPE_FIELD 0x80000u
#define EP_YGO_TYPE_COUNTER 0x100000u
#define EP_YGO_TYPE_FLIP 0x200000u
#define EP_YGO_TYPE_TOON 0x400000u
#define EP_YGO_TYPE_XYZ 0x800000u
#define EP_YGO_TYPE_PENDULUM 0x1000000u
#define EP_YGO_TYPE_SPSUMMON 0x2000000u
#define EP_YGO_TYPE_LINK 0x4000000u
#define EP_YGO_NUM_TYPES 25

/* ATTRIBUTE_* (ygopro.h:563-571 attribute2id: {0, these} -> 0..7) */
#define EP_YGO_ATTRIBUTE_EARTH 0x01
#define EP_YGO_ATTRIBUTE_WATER 0x02
#define EP_YGO_ATTRIBUTE_FIRE 0x04
#define EP_YGO_ATTRIBUTE_WIND 0x08
#define EP_YGO_ATTRIBUTE_LIGHT 0x10
#define EP_YGO_ATTRIBUTE_DARK 0x20
#define EP_YGO_ATTRIBUTE_DEVINE 0x40
#define EP_YGO_NUM_ATTRIBUTES 7

/* RACE_* (ygopro.h:575-604 race2id: {0, bit 0 .. bit 25} -> 0..26), in that order: WARRIOR,
 * SPELLCASTER, FAIRY, FIEND, ZOMBIE, MACHINE, AQUA, PYRO, ROCK, WINDBEAST, PLANT, INSECT, THUNDER,
 * DRAGON, BEAST, BEASTWARRIOR, DINOSAUR, FISH, SEASERPENT, REPTILE, PSYCHO, DEVINE, CREATORGOD,
 * WYRM, CYBERSE, ILLUSION */
#define EP_YGO_RACE(i) (1u << (i)) /* i-th race of the list above */
#define EP_YGO_NUM_RACES 26

/* PHASE_* (ygopro.h:645-659 phase2id: ids 0..9 in this order) */
#define EP_YGO_PHASE_DRAW 0x01
#define EP_YGO_PHASE_STANDBY 0x02
#define EP_YGO_PHASE_MAIN1 0x04
#define EP_YGO_PHASE_BATTLE_START 0x08
#define EP_YGO_PHASE_BATTLE_STEP 0x10
#define EP_YGO_PHASE_DAMAGE 0x20
#define EP_YGO_PHASE_DAMAGE_CAL 0x40
#define EP_YGO_PHASE_BATTLE 0x80
#define EP_YGO_PHASE_MAIN2 0x100
#define EP_YGO_PHASE_END 0x200
#define EP_YGO_NUM_PHASES 10

/* MSG_* of the 14 decisions the tail encodes, in the order of _msgs (ygopro.h:660-669): msg2id = index + 1 */
#define EP_YGO_MSG_SELECT_IDLECMD 11
#define EP_YGO_MSG_SELECT_CHAIN 16
#define EP_YGO_MSG_SELECT_CARD 15
#define EP_YGO_MSG_SELECT_TRIBUTE 20
#define EP_YGO_MSG_SELECT_POSITION 19
#define EP_YGO_MSG_SELECT_EFFECTYN 12
#define EP_YGO_MSG_SELECT_YESNO 13
#define EP_YGO_MSG_SELECT_BATTLECMD 10
#define EP_YGO_MSG_SELECT_UNSELECT_CARD 26
#define EP_YGO_MSG_SELECT_OPTION 14
#define EP_YGO_MSG_SELECT_PLACE 18
#define EP_YGO_MSG_SELECT_SUM 23
#define EP_YGO_MSG_SELECT_DISFIELD 24
#define EP_YGO_MSG_ANNOUNCE_ATTRIB 141
#define EP_YGO_NUM_MSGS 14

/* query_field_card buffer (ygopro.h:2060-2137): the one flag whose presence changes the layout, and
 * the length word of an empty slot */
#define EP_YGO_QUERY_EQUIP_CARD 0x4000u
#define EP_YGO_LEN_EMPTY 4

#ifdef __cplusplus
extern "C" {
#endif

/* The same values as data, for ep_ygo_check_constants: fill it from the real common.h. */
typedef struct {
  int32_t location[8];                     /* DECK, HAND, MZONE, SZONE, GRAVE, REMOVED, EXTRA, OVERLAY */
  int32_t pos[8];                          /* FACEUP_ATTACK, FACEDOWN_ATTACK, FACEUP_DEFENSE, FACEDOWN_DEFENSE, FACEUP, FACEDOWN, ATTACK, DEFENSE */
  uint32_t type[EP_YGO_NUM_TYPES];         /* in the order listed above (type2str key order) */
  uint32_t attribute[EP_YGO_NUM_ATTRIBUTES]; /* EARTH, WATER, FIRE, WIND, LIGHT, DARK, DEVINE */
  uint32_t race[EP_YGO_NUM_RACES];         /* WARRIOR ... ILLUSION, in the order listed above */
  int32_t phase[EP_YGO_NUM_PHASES];        /* DRAW ... END */
  int32_t msg[EP_YGO_NUM_MSGS];            /* in the order of _msgs (ygopro.h:660-669) */
  uint32_t query_equip_card;
  int32_t len_empty;
} ep_ygo_core_constants;

/* Fills `out` with the values this library was built with (the macros above). */
void ep_ygo_default_constants(ep_ygo_core_constants* out);

/* Compares the caller's ygopro-core values with the library's. EP_OK when all agree; otherwise
 * EP_ERR_INVALID, and ep_last_error() names the first constant that differs with both values. No
 * counterpart in the reference (it compiles against common.h itself). */
int ep_ygo_check_constants(const ep_ygo_core_constants* theirs);

#ifdef __cplusplus
}
#endif

/* Structure the arithmetic id tables rely on (see the header comment). */
#if defined(__cplusplus)
static_assert(EP_YGO_LOCATION_DECK == 1 && EP_YGO_LOCATION_HAND == 2 && EP_YGO_LOCATION_MZONE == 4 && EP_YGO_LOCATION_SZONE == 8 &&
              EP_YGO_LOCATION_GRAVE == 16 && EP_YGO_LOCATION_REMOVED == 32 && EP_YGO_LOCATION_EXTRA == 64 &&
              EP_YGO_LOCATION_OVERLAY == 128, "location2id is computed as bit index + 1");
static_assert(EP_YGO_ATTRIBUTE_EARTH == 1 && EP_YGO_ATTRIBUTE_DEVINE == 64, "attribute2id is computed as bit index + 1");
static_assert(EP_YGO_PHASE_DRAW == 1 && EP_YGO_PHASE_END == 0x200, "phase2id is computed as the bit index");
static_assert(EP_YGO_POS_FACEUP == (EP_YGO_POS_FACEUP_ATTACK | EP_YGO_POS_FACEUP_DEFENSE) &&
              EP_YGO_POS_FACEDOWN == (EP_YGO_POS_FACEDOWN_ATTACK | EP_YGO_POS_FACEDOWN_DEFENSE) &&
              EP_YGO_POS_ATTACK == (EP_YGO_POS_FACEUP_ATTACK | EP_YGO_POS_FACEDOWN_ATTACK) &&
              EP_YGO_POS_DEFENSE == (EP_YGO_POS_FACEUP_DEFENSE | EP_YGO_POS_FACEDOWN_DEFENSE), "position2id key set");
#endif

#endif /* ENVPOOL_AMD_YGO_CONSTS_H_ */
