/* flashlight/lib/text/Defines.h -- facade of the MI355X-native decoder. */
#pragma once
#define FL_TEXT_API __attribute__((visibility("default")))
